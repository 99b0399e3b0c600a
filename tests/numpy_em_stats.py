"""Test-only numpy implementation of the local EM statistics (the interface of
kwiiyatta_amd.converter.gmm_fit.HipStats), restating sklearn's formulas
(sklearn/mixture/_gaussian_mixture.py: _estimate_log_gaussian_prob,
_estimate_gaussian_parameters).  Used as the parity reference of the HIP
kernels and to drive the distributed EM loop on CPU (gloo)."""
import numpy as np
from scipy.special import logsumexp


class NumpyStats:
    def __init__(self, X, n_components):
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.n, self.D = self.X.shape
        self.M = n_components
        self.resp = np.zeros((self.n, self.M))
        self.weights = self.means = self.covs = None

    def set_resp_from_labels(self, labels):
        self.resp[:] = 0
        self.resp[np.arange(self.n), labels] = 1

    def set_params(self, weights, means, covs):
        self.weights, self.means, self.covs = weights.copy(), means.copy(), covs.copy()

    def estep(self):
        D = self.D
        wlp = np.empty((self.n, self.M))
        for m in range(self.M):
            L = np.linalg.cholesky(self.covs[m])
            z = np.linalg.solve(L, (self.X - self.means[m]).T)
            wlp[:, m] = (-0.5 * (D * np.log(2 * np.pi) + (z ** 2).sum(0)) - np.log(np.diag(L)).sum()
                         + np.log(self.weights[m]))
        lse = logsumexp(wlp, axis=1)
        self.resp = np.exp(wlp - lse[:, None])
        return float(lse.sum())

    def sums(self):
        return np.hstack((self.resp.sum(0)[:, None], self.resp.T @ self.X))

    def means_from(self, stats):
        self.means = stats[:, 1:] / (stats[:, :1] + 10 * np.finfo(np.float64).eps)

    def cov(self):
        out = np.empty((self.M, self.D, self.D))
        for m in range(self.M):
            diff = self.X - self.means[m]
            out[m] = np.dot(self.resp[:, m] * diff.T, diff)
        return out

    def finalize(self, stats, sxx, reg_covar):
        nk = stats[:, 0] + 10 * np.finfo(np.float64).eps
        self.weights = nk / nk.sum()
        self.covs = sxx / nk[:, None, None]
        self.covs[:, np.arange(self.D), np.arange(self.D)] += reg_covar

    def get_params(self):
        return self.weights, self.means, self.covs
