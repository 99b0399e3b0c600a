"""Full-covariance Gaussian-mixture EM on the GPU, data-parallel over frames.

Stands in for ``sklearn.mixture.GaussianMixture(...).fit(X)`` as the reference
uses it (/root/reference/kwiiyatta/converter/gmm.py:14-26): same hyper-parameters
(``n_components, max_iter, tol, reg_covar, random_state``), same initialisation
(one k-means run, hard assignments -> first M-step), same EM loop and stopping
rule, same fitted attributes (``weights_, means_, covariances_, converged_,
n_iter_, lower_bound_``), which is all that MLPG consumes afterwards.

Every rank holds a shard of the frames.  Per iteration each rank computes the
E-step and the local sufficient statistics with the HIP kernels
(kwy_gmm_em_*), and the statistics -- M*(1+D) doubles, then M*D*D doubles
(10.6 MB for M=64, D=144) -- are summed over ranks with ``all_reduce``
(RCCL over xGMI on GPUs; gloo in the CPU tests).  The M-step itself is
replicated.  This is the only collective of the whole hot path.

The numerical work is delegated to a ``stats`` object with three methods
(estep / sums / cov); ``HipStats`` is the product implementation, the tests
inject a numpy one to exercise the distributed driver without a GPU.
"""
import numpy as np


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def _all_reduce_sum(arr):
    """In-place SUM all-reduce of a numpy array or torch tensor (no-op without a process group)."""
    dist = _dist()
    if dist is None:
        return arr
    import torch
    if isinstance(arr, np.ndarray):
        t = torch.from_numpy(arr)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return arr
    dist.all_reduce(arr, op=dist.ReduceOp.SUM)
    return arr


class HipStats:
    """Local E-step and sufficient statistics on one GPU (device-resident shard)."""

    def __init__(self, X, n_components, device_index=0, ctx=None):
        import torch
        from .. import _lib
        self.torch, self._lib = torch, _lib
        self.dev = torch.device('cuda', device_index)
        self.ctx = ctx or _lib.Context(device_index, stream=torch.cuda.current_stream(self.dev).cuda_stream)
        X = np.ascontiguousarray(X, dtype=np.float64)
        self.n, self.D = X.shape
        self.M = int(n_components)
        self.X = torch.from_numpy(X).to(self.dev)
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.resp = torch.empty((self.n, self.M), **f64)
        self.ll = torch.empty((self.n + 255) // 256, **f64)
        self.status = torch.zeros(4, dtype=torch.int32, device=self.dev)
        self.stats = torch.empty((self.M, self.D + 1), **f64)
        self.sxx = torch.empty((self.M, self.D, self.D), **f64)
        self.means = torch.empty((self.M, self.D), **f64)
        self.weights = torch.empty(self.M, **f64)
        self.covs = torch.empty((self.M, self.D, self.D), **f64)

    def _p(self, t):
        return self._lib.c_vp(t.data_ptr())

    def _chk(self, rc):
        self._lib.check(self.ctx, rc)

    def set_resp_from_labels(self, labels):
        self.resp.zero_()
        idx = self.torch.from_numpy(np.asarray(labels, dtype=np.int64)).to(self.dev)
        self.resp[self.torch.arange(self.n, device=self.dev), idx] = 1.0

    def set_params(self, weights, means, covs):
        for dst, src in ((self.weights, weights), (self.means, means), (self.covs, covs)):
            dst.copy_(self.torch.from_numpy(np.ascontiguousarray(src, dtype=np.float64)))

    def estep(self):
        """responsibilities (kept on the device) and the local sum of log-likelihoods."""
        lib = self._lib.lib
        self._chk(lib.kwy_gmm_em_estep_dev(self.ctx.handle, self._p(self.X), self.n, self.D, self.M,
                                           self._p(self.weights), self._p(self.means), self._p(self.covs),
                                           self._p(self.resp), self._p(self.ll), self._p(self.status)))
        self.ctx.sync()
        if int(self.status[0].item()) != 0:
            raise ValueError('Fitting the mixture model failed because some components have '
                             'ill-defined empirical covariance; increase reg_covar')
        return float(self.ll.sum().item())

    def sums(self):
        """(M, 1+D) local [sum r, sum r x] as a device tensor."""
        self._chk(self._lib.lib.kwy_gmm_em_sums_dev(self.ctx.handle, self._p(self.X), self.n, self.D, self.M,
                                                    self._p(self.resp), self._p(self.stats)))
        self.ctx.sync()
        return self.stats

    def means_from(self, stats):
        self._chk(self._lib.lib.kwy_gmm_em_means_dev(self.ctx.handle, self._p(stats), self.D, self.M,
                                                     self._p(self.means)))

    def cov(self):
        """(M, D, D) local sum r (x - mu)(x - mu)' around self.means, device tensor."""
        self._chk(self._lib.lib.kwy_gmm_em_cov_dev(self.ctx.handle, self._p(self.X), self.n, self.D, self.M,
                                                   self._p(self.resp), self._p(self.means), self._p(self.sxx)))
        self.ctx.sync()
        return self.sxx

    def finalize(self, stats, sxx, reg_covar):
        self._chk(self._lib.lib.kwy_gmm_em_finalize_dev(self.ctx.handle, self._p(stats), self._p(sxx), self.D,
                                                        self.M, float(reg_covar), self._p(self.weights),
                                                        self._p(self.covs)))
        self.ctx.sync()

    def get_params(self):
        return (self.weights.cpu().numpy(), self.means.cpu().numpy(), self.covs.cpu().numpy())


def em_fit(stats, n_total, max_iter=100, tol=1e-3, reg_covar=1e-6, verbose=0):
    """The EM loop of sklearn's BaseMixture.fit_predict, starting from the
    responsibilities already stored in `stats` (hard k-means assignments).
    `stats` computes on the local shard; sufficient statistics are all-reduced."""
    def m_step():
        s = _all_reduce_sum(stats.sums())
        stats.means_from(s)
        c = _all_reduce_sum(stats.cov())
        stats.finalize(s, c, reg_covar)

    m_step()                                    # initialisation from the hard assignments
    lower_bound, converged, n_iter = -np.inf, False, 0
    for n_iter in range(1, max_iter + 1):
        prev = lower_bound
        ll = np.array([stats.estep()])
        _all_reduce_sum(ll)
        m_step()
        lower_bound = float(ll[0]) / n_total
        change = lower_bound - prev
        if verbose:
            print(f'  Iteration {n_iter}\t lower bound {lower_bound:.6f}\t change {change:.6f}')
        if abs(change) < tol:
            converged = True
            break
    return lower_bound, converged, n_iter


class GaussianMixtureHIP:
    """Drop-in for the sklearn GaussianMixture object held by GMMFeatureConverter
    (full covariance, one initialisation)."""
    covariance_type = 'full'

    def __init__(self, n_components=1, covariance_type='full', tol=1e-3, reg_covar=1e-6, max_iter=100,
                 n_init=1, init_params='kmeans', random_state=None, verbose=0, device_index=0, **unused):
        if covariance_type != 'full' or n_init != 1 or init_params != 'kmeans':
            raise NotImplementedError('GaussianMixtureHIP implements the configuration the reference uses: '
                                      "covariance_type='full', n_init=1, init_params='kmeans'")
        self.n_components, self.tol, self.reg_covar, self.max_iter = n_components, tol, reg_covar, max_iter
        self.random_state, self.verbose, self.device_index = random_state, verbose, device_index

    def _initial_labels(self, X):
        """sklearn's initialisation: one KMeans run with the mixture's random state.
        With several ranks, rank 0's centres are broadcast and every rank labels its
        own shard by the nearest centre."""
        from sklearn.cluster import KMeans
        from sklearn.utils import check_random_state
        dist = _dist()
        rs = check_random_state(self.random_state)
        if dist is None:
            return KMeans(n_clusters=self.n_components, n_init=1, random_state=rs).fit(X).labels_
        import torch
        centres = np.zeros((self.n_components, X.shape[1]))
        if dist.get_rank() == 0:
            centres[:] = KMeans(n_clusters=self.n_components, n_init=1, random_state=rs).fit(X).cluster_centers_
        t = torch.from_numpy(centres)
        if dist.get_backend() == 'nccl':
            t = t.cuda()
            dist.broadcast(t, 0)
            centres = t.cpu().numpy()
        else:
            dist.broadcast(t, 0)
        d2 = (X ** 2).sum(1)[:, None] - 2 * X @ centres.T + (centres ** 2).sum(1)[None, :]
        return d2.argmin(1)

    def fit(self, X, y=None, stats=None):
        X = np.ascontiguousarray(X, dtype=np.float64)
        if X.shape[0] < self.n_components:
            raise ValueError(f'Expected n_samples >= n_components but got n_components = {self.n_components}, '
                             f'n_samples = {X.shape[0]}')
        if stats is None:
            stats = HipStats(X, self.n_components, device_index=self.device_index)
        stats.set_resp_from_labels(self._initial_labels(X))
        n_total = np.array([float(X.shape[0])])
        _all_reduce_sum(n_total)
        if self.verbose:
            print('Initialization 0')
        self.lower_bound_, self.converged_, self.n_iter_ = em_fit(
            stats, float(n_total[0]), self.max_iter, self.tol, self.reg_covar, self.verbose > 1)
        if self.verbose:
            print(f'Initialization converged: {self.converged_}')
        self.weights_, self.means_, self.covariances_ = stats.get_params()
        return self
