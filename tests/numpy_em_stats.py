"""Test-only numpy implementation of the local fit statistics (the interface of
kwiiyatta_amd.converter.gmm_fit.HipStats), restating sklearn's formulas
(sklearn/mixture/_gaussian_mixture.py: _estimate_log_gaussian_prob, _estimate_gaussian_parameters;
sklearn/cluster/_kmeans.py: _kmeans_plusplus, lloyd_iter_chunked_dense).  Used as the parity reference of
the HIP kernels and to drive the distributed k-means / EM driver on CPU (gloo).  Tensors handed to the
driver are CPU torch tensors that share memory with the numpy arrays."""
import numpy as np
import torch
from scipy.special import logsumexp


def _t(a):
    return torch.from_numpy(a)


class NumpyStats:
    torch = torch
    dev = torch.device('cpu')

    def __init__(self, X, n_components):
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.n, self.D = self.X.shape
        self.M = n_components
        self.resp = np.zeros((self.n, self.M))
        self.weights = self.means = self.covs = None
        self.failed = 0.0

    def scope(self):
        import contextlib
        return contextlib.nullcontext()

    # ---- EM ---------------------------------------------------------------------------------------
    def set_resp_from_labels(self, labels):
        self.resp[:] = 0
        self.resp[np.arange(self.n), labels] = 1

    def set_params(self, weights, means, covs):
        self.weights, self.means, self.covs = weights.copy(), means.copy(), covs.copy()

    def estep(self):
        D = self.D
        wlp = np.empty((self.n, self.M))
        self.failed = 0.0
        for m in range(self.M):
            try:
                L = np.linalg.cholesky(self.covs[m])
            except np.linalg.LinAlgError:
                self.failed = 1.0
                return _t(np.zeros(1))
            z = np.linalg.solve(L, (self.X - self.means[m]).T)
            wlp[:, m] = (-0.5 * (D * np.log(2 * np.pi) + (z ** 2).sum(0)) - np.log(np.diag(L)).sum()
                         + np.log(self.weights[m]))
        lse = logsumexp(wlp, axis=1)
        self.resp = np.exp(wlp - lse[:, None])
        return _t(np.array([lse.sum()]))

    def estep_failed(self):
        return _t(np.array([self.failed]))

    def sums(self):
        return _t(np.hstack((self.resp.sum(0)[:, None], self.resp.T @ self.X)))

    def means_from(self, stats):
        stats = stats.numpy()
        self.means = stats[:, 1:] / (stats[:, :1] + 10 * np.finfo(np.float64).eps)

    def cov(self, stats=None):
        out = np.empty((self.M, self.D, self.D))
        for m in range(self.M):
            diff = self.X - self.means[m]
            out[m] = np.dot(self.resp[:, m] * diff.T, diff)
        return _t(out)

    def finalize(self, stats, sxx, reg_covar):
        stats, sxx = stats.numpy(), sxx.numpy()
        nk = stats[:, 0] + 10 * np.finfo(np.float64).eps
        self.weights = nk / nk.sum()
        self.covs = sxx / nk[:, None, None]
        self.covs[:, np.arange(self.D), np.arange(self.D)] += reg_covar

    def get_params(self):
        return self.weights, self.means, self.covs

    # ---- k-means ------------------------------------------------------------------------------------
    def km_colstats(self, shift=None):
        x = self.X if shift is None else self.X - shift.numpy()
        return _t(np.stack((x.sum(0), (x * x).sum(0))))

    def km_begin(self, mean):
        self.Xc = self.X - mean.numpy()
        self.xsq = np.einsum('ij,ij->i', self.Xc, self.Xc)
        self.closest = np.zeros(self.n)
        self.newd = None
        self.labels = np.full(self.n, -1, dtype=np.int32)

    def km_end(self):
        self.Xc = self.xsq = self.closest = self.newd = None

    def km_row(self, i):
        return _t(self.Xc[i.numpy()])

    def km_closest_total(self):
        return _t(np.array([self.closest.sum()]))

    def km_pick(self, lo, hi, vals, first, last):
        cum = np.cumsum(self.closest)
        out = np.empty(len(vals), dtype=np.int64)
        for c, v in enumerate(vals.numpy()):
            mine = (first or v > float(lo)) and (v <= float(hi) or last)
            out[c] = min(int(np.searchsorted(cum, v - float(lo))), self.n - 1) if mine else -1
        return _t(out)

    def km_candidates(self, cand, use_closest):
        y = cand.numpy()
        d = -2.0 * (y @ self.Xc.T)
        d += np.einsum('ij,ij->i', y, y)[:, None]
        d += self.xsq[None, :]
        np.maximum(d, 0, out=d)
        if use_closest:
            np.minimum(self.closest, d, out=d)
        self.newd = d
        return _t(d.sum(1))

    def km_accept(self, best):
        self.closest = self.newd[int(best)].copy()

    def km_assign(self, centers):
        c = centers.numpy()
        d = (c * c).sum(1)[None, :] - 2.0 * (self.Xc @ c.T)
        new = d.argmin(1).astype(np.int32)
        changed = int((new != self.labels).sum())
        self.labels = new
        self.set_resp_from_labels(new)
        return _t(np.array([changed], dtype=np.int64))

    def km_sums(self):
        return _t(np.hstack((self.resp.sum(0)[:, None], self.resp.T @ self.Xc)))

    def km_update(self, stats, centers_old, centers_new):
        st, old, new = stats.numpy(), centers_old.numpy(), centers_new.numpy()
        has = st[:, 0] > 0
        new[:] = old
        new[has] = st[has, 1:] * (1.0 / st[has, :1])
        return _t(((new - old) ** 2).sum(1))

    def km_labels_of(self, i):
        return _t(self.labels[i.numpy()])

    def km_far_rows(self, centers, k):
        d = ((self.Xc - centers.numpy()[self.labels]) ** 2).sum(1)
        i = np.argsort(-d, kind='stable')[:k]
        return _t(d[i]), _t(i.astype(np.int64))
