"""Full-covariance Gaussian-mixture fit on the GPU, data-parallel over frames.

Stands in for ``sklearn.mixture.GaussianMixture(...).fit(X)`` as the reference
uses it (/root/reference/kwiiyatta/converter/gmm.py:14-26): same hyper-parameters
(``n_components, max_iter, tol, reg_covar, random_state``), same initialisation
(one k-means run -- k-means++ seeding, Lloyd iterations -- whose hard assignments
feed the first M-step), same EM loop and stopping rule, same fitted attributes
(``weights_, means_, covariances_, converged_, n_iter_, lower_bound_``), which is
all that MLPG consumes afterwards.

Every rank holds a shard of the frames; the global row order is rank 0's rows, then
rank 1's, ...  Both phases compute on the local shard with the HIP kernels
(kwy_km_*, kwy_gmm_em_*) and exchange only small quantities:

  k-means++   per centre: shard totals of the closest distances (all_gather), the
              <= 8 candidate rows and their potentials (all_reduce)
  Lloyd       per iteration: centroid sums and counts, M (D+1) doubles, and the number
              of changed labels (all_reduce)
  EM          per iteration: M (1+D) doubles, then M D D doubles (10.6 MB for M = 64,
              D = 144), and the log-likelihood (all_reduce)

over ``torch.distributed`` (RCCL over xGMI on GPUs: every reduced buffer is a device
tensor; gloo in the CPU tests).  The random draws of the seeding come from the
mixture's ``random_state`` on the host exactly as scikit-learn consumes them, and
every decision is taken on globally reduced numbers, so the result does not depend on
the number of ranks beyond the rounding of the sums.

The numerical work is delegated to a ``stats`` object; ``HipStats`` is the product
implementation, the tests inject a numpy one to exercise the driver without a GPU.
"""
import numpy as np


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


class Comm:
    """SUM all-reduce / all-gather of torch tensors living on the statistics' device.  RCCL takes
    device tensors as they are; gloo (CPU tests) gets a host copy.  Without a process group every
    operation is the identity."""

    def __init__(self):
        self.dist = _dist()
        self.rank = self.dist.get_rank() if self.dist else 0
        self.world = self.dist.get_world_size() if self.dist else 1
        self.device_native = bool(self.dist) and self.dist.get_backend() == 'nccl'

    def all_reduce(self, t):
        """in place; returns t"""
        if self.dist is None:
            return t
        if t.is_cuda and not self.device_native:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            if not t.is_cuda and self.device_native:
                raise RuntimeError('RCCL reduces device tensors only: keep the statistics on the GPU')
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t

    def all_gather_scalar(self, t):
        """t: one-element tensor -> (world,) tensor on the same device, in rank order"""
        import torch
        if self.dist is None:
            return t.reshape(1).clone()
        out = torch.zeros(self.world, dtype=t.dtype, device=t.device)
        out[self.rank] = t.reshape(())
        return self.all_reduce(out)


class HipStats:
    """Local statistics on one GPU: the device-resident shard and the HIP kernels over it."""

    def __init__(self, X, n_components, device_index=0, ctx=None):
        import torch
        from .. import _lib
        self.torch, self._lib = torch, _lib
        self.dev = torch.device('cuda', device_index)
        # One stream for everything the fit does: the library's kernels, the driver's small torch ops and
        # the collectives (torch.distributed orders RCCL work after the current stream).  `scope()` makes it
        # torch's current stream; the fit entry points run inside it.
        if ctx is None:
            self.stream = torch.cuda.Stream(device=self.dev)
            self.ctx = _lib.Context(device_index, stream=self.stream.cuda_stream)
        else:
            self.ctx = ctx
            self.stream = torch.cuda.ExternalStream(_lib.lib.kwy_ctx_stream(ctx.handle), device=self.dev)
        with self.scope():
            self._allocate(X, n_components)

    def scope(self):
        return self.torch.cuda.stream(self.stream)

    def _allocate(self, X, n_components):
        torch = self.torch
        if isinstance(X, torch.Tensor):         # a shard already in HBM (the corpus driver writes it there)
            if X.dtype != torch.float64 or not X.is_contiguous() or X.device != self.dev:
                raise ValueError('HipStats: X must be a contiguous float64 tensor on the fit device')
            self.X = X
        else:
            self.X = torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(self.dev)
        self.n, self.D = self.X.shape
        self.M = int(n_components)
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
        return self._lib.c_vp(t.data_ptr()) if t is not None else None

    def _chk(self, rc):
        self._lib.check(self.ctx, rc)

    def tensor(self, a, dtype=None):
        t = self.torch.as_tensor(a, dtype=dtype)
        return t.to(self.dev)

    # ---- EM -----------------------------------------------------------------------------------
    def set_resp_from_labels(self, labels):
        self.resp.zero_()
        idx = self.torch.from_numpy(np.asarray(labels, dtype=np.int64)).to(self.dev)
        self.resp[self.torch.arange(self.n, device=self.dev), idx] = 1.0

    def sort_rows_by_component(self):
        """Reorder the local rows so that the frames of a component sit together (stable sort by the component of
        largest responsibility: right after k-means, its cluster).  The fit does not depend on the order of its rows
        beyond the rounding of its sums, and the covariance kernel skips whole tiles of frames that carry no weight
        for a component (kwy_gmm_em_cov_stats_dev): with the rows grouped that is nearly all of them.  The caller's
        matrix is left alone (a sorted copy takes its place here)."""
        torch = self.torch
        order = torch.argsort(self.resp.argmax(dim=1), stable=True)
        self.X = self.X.index_select(0, order)
        self.resp = self.resp.index_select(0, order)

    def set_params(self, weights, means, covs):
        for dst, src in ((self.weights, weights), (self.means, means), (self.covs, covs)):
            dst.copy_(self.torch.from_numpy(np.ascontiguousarray(src, dtype=np.float64)))

    def estep(self):
        """responsibilities (kept on the device); returns the local sum of log-likelihoods as a
        one-element device tensor"""
        lib = self._lib.lib
        self._chk(lib.kwy_gmm_em_estep_dev(self.ctx.handle, self._p(self.X), self.n, self.D, self.M,
                                           self._p(self.weights), self._p(self.means), self._p(self.covs),
                                           self._p(self.resp), self._p(self.ll), self._p(self.status)))
        return self.ll.sum().reshape(1)

    def estep_failed(self):
        """one-element device tensor: nonzero if a covariance was not positive definite"""
        return self.status[:1].to(self.torch.float64)

    def sums(self):
        """(M, 1+D) local [sum r, sum r x] as a device tensor."""
        self._chk(self._lib.lib.kwy_gmm_em_sums_dev(self.ctx.handle, self._p(self.X), self.n, self.D, self.M,
                                                    self._p(self.resp), self._p(self.stats)))
        return self.stats

    def means_from(self, stats):
        self._chk(self._lib.lib.kwy_gmm_em_means_dev(self.ctx.handle, self._p(stats), self.D, self.M,
                                                     self._p(self.means)))

    def cov(self, stats=None):
        """(M, D, D) local sum r (x - mu)(x - mu)' around self.means, device tensor.  stats: the (globally reduced)
        sums of `sums()` -- frames below 2^-70 of a mixture's mass are then left out of its sum (kwy_gmm_em_cov_stats_dev)."""
        self._chk(self._lib.lib.kwy_gmm_em_cov_stats_dev(self.ctx.handle, self._p(self.X), self.n, self.D, self.M,
                                                         self._p(self.resp), self._p(self.means),
                                                         self._p(stats) if stats is not None else None, self._p(self.sxx)))
        return self.sxx

    def finalize(self, stats, sxx, reg_covar):
        self._chk(self._lib.lib.kwy_gmm_em_finalize_dev(self.ctx.handle, self._p(stats), self._p(sxx), self.D,
                                                        self.M, float(reg_covar), self._p(self.weights),
                                                        self._p(self.covs)))

    def get_params(self):
        with self.scope():
            out = (self.weights.cpu().numpy(), self.means.cpu().numpy(), self.covs.cpu().numpy())
        self.ctx.sync()
        return out

    # ---- k-means --------------------------------------------------------------------------------
    def km_colstats(self, shift=None):
        """(2, D): column sums and sums of squares of X - shift over the shard"""
        out = self.torch.empty((2, self.D), dtype=self.torch.float64, device=self.dev)
        self._chk(self._lib.lib.kwy_km_colstats_dev(self.ctx.handle, self._p(self.X), self.n, self.D,
                                                    self._p(shift), self._p(out)))
        return out

    def km_begin(self, mean):
        """centred copy of the shard and its row norms; buffers of the seeding and of Lloyd's iterations"""
        t = self.torch
        f64 = dict(dtype=t.float64, device=self.dev)
        self.Xc = t.empty_like(self.X)
        self.xsq = t.empty(self.n, **f64)
        self._chk(self._lib.lib.kwy_km_center_dev(self.ctx.handle, self._p(self.X), self.n, self.D, self._p(mean),
                                                  self._p(self.Xc), self._p(self.xsq)))
        self.closest = t.empty(self.n, **f64)
        self.newd = t.empty((8, self.n), **f64)
        self.pots = t.zeros(8, **f64)
        self.csums = t.empty(int(self._lib.lib.kwy_km_chunks(self.n)), **f64)
        self.total = t.zeros(1, **f64)
        self.pick = t.empty(8, dtype=t.int64, device=self.dev)
        self.labels = t.full((self.n,), -1, dtype=t.int32, device=self.dev)
        self.changed = t.zeros(1, dtype=t.int64, device=self.dev)
        self.shift2 = t.empty(self.M, **f64)

    def km_end(self):
        self.ctx.sync()      # nothing in flight still reads the buffers that go back to the allocator
        for name in ('Xc', 'xsq', 'closest', 'newd', 'csums'):
            setattr(self, name, None)

    def km_row(self, i):
        return self.Xc[i]

    def km_closest_total(self):
        self._chk(self._lib.lib.kwy_km_pp_total_dev(self.ctx.handle, self._p(self.closest), self.n,
                                                    self._p(self.csums), self._p(self.total)))
        return self.total

    def km_pick(self, lo, hi, vals, first, last):
        """local row indices (int64 device tensor, -1: not in this shard) of searchsorted(lo + cumsum(closest), vals);
        this shard owns the values in (lo, hi] (first / last shard: also those below / beyond)"""
        L = len(vals)
        self._chk(self._lib.lib.kwy_km_pp_pick_dev(self.ctx.handle, self._p(self.closest), self.n, self._p(self.csums),
                                                   self._p(lo), self._p(hi), self._p(vals), L, int(first), int(last),
                                                   self._p(self.pick)))
        return self.pick[:L]

    def km_candidates(self, cand, use_closest):
        """newd (L, n) and the local potentials (L,) of the candidate rows `cand` (L, D)"""
        L = cand.shape[0]
        self._chk(self._lib.lib.kwy_km_pp_dist_dev(self.ctx.handle, self._p(self.Xc), self._p(self.xsq), self.n, self.D,
                                                   self._p(cand), L, self._p(self.closest) if use_closest else None,
                                                   self._p(self.newd), self._p(self.pots)))
        return self.pots[:L]

    def km_accept(self, best):
        """closest <- newd[best]; best: one-element int64 device tensor"""
        self.torch.index_select(self.newd, 0, best.reshape(1), out=self.closest.reshape(1, -1))

    def km_assign(self, centers):
        """labels of the nearest centres (one-hot into resp); returns the local number of changed labels
        (one-element int64 device tensor)"""
        self._chk(self._lib.lib.kwy_km_assign_dev(self.ctx.handle, self._p(self.Xc), self.n, self.D, self._p(centers),
                                                  self.M, self._p(self.labels), self._p(self.resp),
                                                  self._p(self.changed)))
        return self.changed

    def km_sums(self):
        """(M, 1+D) local [count, sum of centred rows] per label"""
        self._chk(self._lib.lib.kwy_gmm_em_sums_dev(self.ctx.handle, self._p(self.Xc), self.n, self.D, self.M,
                                                    self._p(self.resp), self._p(self.stats)))
        return self.stats

    def km_update(self, stats, centers_old, centers_new):
        self._chk(self._lib.lib.kwy_km_update_dev(self.ctx.handle, self._p(stats), self._p(centers_old), self.M, self.D,
                                                  self._p(centers_new), self._p(self.shift2)))
        return self.shift2

    def km_lloyd(self, centers2, abs_tol, iterations, max_iter, state, log):
        """`iterations` Lloyd iterations enqueued back to back, the stopping decision taken on the device
        (kwy_km_lloyd_dev; one shard only: the sums are not reduced between ranks)"""
        self._chk(self._lib.lib.kwy_km_lloyd_dev(self.ctx.handle, self._p(self.Xc), self.n, self.D, self._p(centers2),
                                                 self.M, self._p(self.labels), self._p(self.resp),
                                                 self._p(self.changed), self._p(self.stats), self._p(self.shift2),
                                                 float(abs_tol), int(iterations), int(max_iter), self._p(state),
                                                 self._p(log)))

    def km_onehot(self):
        """resp <- the one-hot rows of the labels (the device loop does not keep them up to date)"""
        self._chk(self._lib.lib.kwy_km_onehot_dev(self.ctx.handle, self._p(self.labels), self.n, self.M,
                                                  self._p(self.resp)))

    def km_labels_of(self, i):
        return self.labels[i]

    def km_far_rows(self, centers, k):
        """the k rows farthest from their own centre: (distances, local indices), descending (empty-cluster relocation)"""
        t = self.torch
        d = (self.Xc - centers[self.labels.long()]).pow(2).sum(1)
        v, i = t.topk(d, min(k, self.n))
        return v, i


LLOYD_BATCH = 16          # Lloyd iterations enqueued per read-back of the device's stopping decision (one shard)


def kmeans_init(stats, n_components, random_state, max_iter=300, tol=1e-4, verbose=0):
    with stats.scope():
        return _kmeans_init(stats, n_components, random_state, max_iter, tol, verbose)


def _kmeans_init(stats, n_components, random_state, max_iter, tol, verbose):
    """sklearn.cluster.KMeans(n_clusters=M, n_init=1, random_state=rs).fit(X).labels_ on the sharded rows
    (sklearn/cluster/_kmeans.py: fit, _kmeans_plusplus, _kmeans_single_lloyd), left in `stats` as one-hot
    responsibilities.  Returns (n_iter, centres as numpy, in the centred coordinates + the mean)."""
    import torch
    from sklearn.utils import check_random_state
    comm = Comm()
    rs = check_random_state(random_state)
    M = int(n_components)
    dev = stats.dev
    f64 = dict(dtype=torch.float64, device=dev)
    n_local = stats.n
    counts = comm.all_gather_scalar(torch.tensor([float(n_local)], **f64))
    n_total = int(round(float(counts.sum().item())))
    row0 = int(round(float(counts[:comm.rank].sum().item())))      # global index of this shard's first row
    if n_total < M:
        raise ValueError(f'n_samples={n_total} should be >= n_clusters={M}.')

    # KMeans.fit: centre the data, tolerance relative to the mean variance
    s = comm.all_reduce(stats.km_colstats(None))
    mean = s[0] / n_total
    stats.km_begin(mean)
    s2 = comm.all_reduce(stats.km_colstats(mean))
    abs_tol = float((s2[1] / n_total - (s2[0] / n_total) ** 2).mean().item()) * tol

    # the random numbers of _kmeans_plusplus, in its order: one for random_state.choice, then
    # uniform(size=n_local_trials) per further centre
    n_trials = 2 + int(np.log(M))
    if n_trials > 8:
        raise ValueError('k-means++ with more than 8 candidates per centre (n_components > 403) is not supported')
    cdf = np.full(n_total, 1.0 / n_total).cumsum()
    cdf /= cdf[-1]
    first_id = int(cdf.searchsorted(rs.random_sample(), side='right'))
    u = torch.from_numpy(rs.uniform(size=(M - 1, n_trials)) if M > 1 else np.zeros((0, n_trials))).to(dev)

    def fetch_rows(idx):
        """rows of the global matrix by local index (-1: another shard's): (L, D) on every rank"""
        L = idx.shape[0]
        buf = torch.zeros((L, stats.D), **f64)
        mine = idx >= 0
        safe = idx.clamp(0, n_local - 1)
        buf.copy_(stats.km_row(safe) * mine.to(torch.float64)[:, None])
        return comm.all_reduce(buf)

    centers = torch.zeros((M, stats.D), **f64)
    local_first = first_id - row0
    idx0 = torch.tensor([local_first if 0 <= local_first < n_local else -1], dtype=torch.int64, device=dev)
    centers[0] = fetch_rows(idx0)[0]
    pot = comm.all_reduce(stats.km_candidates(centers[:1].contiguous(), use_closest=False).clone())
    stats.km_accept(torch.zeros(1, dtype=torch.int64, device=dev))
    current_pot = pot[0]
    for c in range(1, M):
        # shard r owns the values in (bounds[r], bounds[r + 1]]: one cumulative sum of the gathered totals, the same
        # numbers on every rank, so that a value on a boundary has exactly one owner
        totals = comm.all_gather_scalar(stats.km_closest_total())
        bounds = torch.cat((torch.zeros(1, **f64), torch.cumsum(totals, 0)))
        lo, hi = bounds[comm.rank:comm.rank + 1].contiguous(), bounds[comm.rank + 1:comm.rank + 2].contiguous()
        vals = (u[c - 1] * current_pot).contiguous()
        idx = stats.km_pick(lo, hi, vals, first=comm.rank == 0, last=comm.rank == comm.world - 1)
        cand = fetch_rows(idx)
        pots = comm.all_reduce(stats.km_candidates(cand, use_closest=True).clone())
        best = torch.argmin(pots).reshape(1)
        stats.km_accept(best)
        current_pot = pots[best[0]]
        centers[c] = cand[best[0]]

    # _kmeans_single_lloyd
    strict = False
    n_iter = 0
    if comm.world == 1 and hasattr(stats, 'km_lloyd'):
        # One shard: the device takes the stopping decision after every iteration and the host reads it once per
        # batch of LLOYD_BATCH iterations (the iterations enqueued behind the last one do nothing) -- a host round
        # trip per iteration was 40 % of the loop at 4.4e5 x 144.  An empty cluster (sklearn relocates it BEFORE the
        # update) hands the iteration back after its sums.
        c2 = torch.stack((centers, torch.zeros_like(centers))).contiguous()
        state = torch.zeros(4, dtype=torch.int64, device=dev)
        log = torch.zeros((max_iter, 2), **f64)
        seen = 0
        while True:
            stats.km_lloyd(c2, abs_tol, min(LLOYD_BATCH, max_iter - n_iter), max_iter, state, log)
            code, n_iter, n_changed = state.tolist()[:3]
            if code == 3:
                stats.km_onehot()
                st = stats.stats
                cur = n_iter & 1
                _relocate_empty_clusters(stats, comm, st, c2[cur], (st[:, 0] == 0).nonzero().flatten(), row0)
                shift_tot = float(stats.km_update(st, c2[cur], c2[cur ^ 1]).sum().item())
                n_iter += 1
                log[n_iter - 1, 0], log[n_iter - 1, 1] = float(n_changed), shift_tot
                code = 1 if n_changed == 0 else 2 if shift_tot <= abs_tol else 4 if n_iter >= max_iter else 0
                state.copy_(torch.tensor([code, n_iter, n_changed, 0], dtype=torch.int64))
            if verbose:
                for i, (ch, sh) in enumerate(log[seen:n_iter].tolist(), seen + 1):
                    print(f'  k-means iteration {i}: {int(ch)} labels changed, centre shift {sh:.3e}')
            seen = n_iter
            if code:
                strict = code == 1
                break
        centers = c2[n_iter & 1].clone()
        if strict:
            stats.km_onehot()               # (not strict: the assignment below writes them)
    else:
        centers_new = torch.empty_like(centers)
        for n_iter in range(1, max_iter + 1):
            changed = comm.all_reduce(stats.km_assign(centers).clone())
            st = comm.all_reduce(stats.km_sums())
            # ONE host round trip per iteration: the centres are updated as if no cluster were empty (the usual case)
            # and {empty clusters, changed labels, centre shift} come back together; an empty cluster -- sklearn
            # relocates it BEFORE the update -- makes the iteration redo its update (km_update is a pure function of
            # `st` and the old centres)
            shift2 = stats.km_update(st, centers, centers_new)
            n_empty, n_changed, shift_tot = torch.stack(((st[:, 0] == 0).sum().to(torch.float64),
                                                         changed.reshape(()).to(torch.float64), shift2.sum())).tolist()
            if n_empty > 0:
                _relocate_empty_clusters(stats, comm, st, centers, (st[:, 0] == 0).nonzero().flatten(), row0)
                shift_tot = float(stats.km_update(st, centers, centers_new).sum().item())
            centers, centers_new = centers_new, centers
            n_changed = int(n_changed)
            if verbose:
                print(f'  k-means iteration {n_iter}: {n_changed} labels changed, centre shift {shift_tot:.3e}')
            if n_changed == 0:
                strict = True
                break
            if shift_tot <= abs_tol:
                break
    if not strict:
        stats.km_assign(centers)            # labels that match the final centres
    out = (centers + mean).cpu().numpy()
    stats.km_end()
    return n_iter, out


def _relocate_empty_clusters(stats, comm, st, centers, empty, row0):
    """sklearn's _relocate_empty_clusters_dense on the reduced [count, sums]: every empty cluster takes one of
    the rows farthest from their own centre (largest first), which leaves its old cluster."""
    import torch
    k = len(empty)
    v, i = stats.km_far_rows(centers, k)
    rows = stats.km_row(i)
    labs = stats.km_labels_of(i).to(torch.float64)
    pack = torch.zeros((comm.world, k, stats.D + 3), dtype=torch.float64, device=stats.dev)
    m = len(v)
    pack[comm.rank, :m, 0] = v
    pack[comm.rank, :m, 1] = (i + row0).to(torch.float64)
    pack[comm.rank, :m, 2] = labs
    pack[comm.rank, :m, 3:] = rows
    pack[comm.rank, m:, 0] = -1.0
    comm.all_reduce(pack)
    flat = pack.reshape(-1, stats.D + 3)
    order = torch.argsort(flat[:, 0], descending=True, stable=True)[:k]
    for slot, j in zip(order.tolist(), empty.tolist()):
        row, old = flat[slot, 3:], int(flat[slot, 2].item())
        st[old, 1:] -= row
        st[old, 0] -= 1.0
        st[j, 1:] = row
        st[j, 0] = 1.0


def em_fit(stats, n_total, max_iter=100, tol=1e-3, reg_covar=1e-6, verbose=0):
    with stats.scope():
        return _em_fit(stats, n_total, max_iter, tol, reg_covar, verbose)


def _em_fit(stats, n_total, max_iter, tol, reg_covar, verbose):
    """The EM loop of sklearn's BaseMixture.fit_predict, starting from the responsibilities already
    stored in `stats` (hard k-means assignments).  `stats` computes on the local shard; sufficient
    statistics are all-reduced."""
    comm = Comm()

    def m_step():
        s = comm.all_reduce(stats.sums())
        stats.means_from(s)
        c = comm.all_reduce(stats.cov(s))
        stats.finalize(s, c, reg_covar)

    m_step()                                    # initialisation from the hard assignments
    lower_bound, converged, n_iter = -np.inf, False, 0
    for n_iter in range(1, max_iter + 1):
        prev = lower_bound
        ll = stats.estep()
        both = comm.all_reduce(stats.torch.cat((ll, stats.estep_failed())))
        ll_total, failed = (float(v) for v in both.tolist())
        if failed:
            raise ValueError('Fitting the mixture model failed because some components have '
                             'ill-defined empirical covariance; increase reg_covar')
        m_step()
        lower_bound = ll_total / n_total
        change = lower_bound - prev
        if verbose:
            print(f'  Iteration {n_iter}\t lower bound {lower_bound:.6f}\t change {change:.6f}')
        if abs(change) < tol:
            converged = True
            break
    return lower_bound, converged, n_iter


class _DeviceDoubles:
    """`count` doubles at a raw device address, as something torch.as_tensor can view without copying"""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {'shape': (int(count),), 'typestr': '<f8', 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


def library_comm(device_index=0):
    """torch.distributed's default process group as the library's communicator (`kwy_comm`, include/kwy.h): the
    all-reduce callback views the library's device buffer as a tensor and reduces it with `Comm` on the stream the
    library names -- RCCL under 'nccl', a host round trip under 'gloo'.  Returns (struct, keep-alive objects), or
    (None, ()) without a process group.  (A C or C++ binder hands ncclAllReduce straight to the library instead.)"""
    import ctypes
    import torch
    comm = Comm()
    if comm.dist is None:
        return None, ()
    dev = torch.device('cuda', device_index)
    CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p)

    class KwyComm(ctypes.Structure):
        _fields_ = [('rank', ctypes.c_int), ('world', ctypes.c_int), ('all_reduce_sum', CB), ('user', ctypes.c_void_p)]

    def reduce(user, buf, count, stream):
        try:
            t = torch.as_tensor(_DeviceDoubles(buf, count), device=dev)
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
                comm.all_reduce(t)
            return 0
        except Exception:            # no exception may cross the C frames above
            import traceback
            traceback.print_exc()
            return 1
    cb = CB(reduce)
    return KwyComm(comm.rank, comm.world, cb, None), (cb, comm)


def fit_one_call(X, n_components, max_iter=100, tol=1e-3, reg_covar=1e-6, random_state=0, device_index=0, ctx=None,
                 distributed=False):
    """The fit of one rank's rows through the library's single entry point `kwy_gmm_fit_dev` (the control flow of this
    module in C++, for binders without a Python driver).  X: (n, D) numpy array or float64 device tensor; an integer
    `random_state`.  distributed=True: `kwy_gmm_fit_comm_dev` with torch.distributed's default group as the
    communicator -- X is this rank's shard and every rank returns the model of all rows.
    Returns a GaussianMixtureHIP carrying the fitted attributes."""
    import ctypes
    import torch
    from .. import _lib
    dev = torch.device('cuda', device_index)
    Xd = X if isinstance(X, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(X, dtype=np.float64)).to(dev)
    n, D = Xd.shape
    M = int(n_components)
    ctx = ctx or _lib.Context(device_index)
    g = GaussianMixtureHIP(n_components=M, max_iter=max_iter, tol=tol, reg_covar=reg_covar, random_state=random_state,
                           device_index=device_index)
    g.weights_, g.means_, g.covariances_ = np.empty(M), np.empty((M, D)), np.empty((M, D, D))
    it, conv, kit, lb = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
    torch.cuda.synchronize(dev)
    kc, alive = library_comm(device_index) if distributed else (None, ())
    if kc is None:
        _lib.check(ctx, _lib.lib.kwy_gmm_fit_dev(ctx.handle, _lib.c_vp(Xd.data_ptr()), n, D, M, int(max_iter), float(tol),
                                                 float(reg_covar), int(random_state), _lib.ptr(g.weights_),
                                                 _lib.ptr(g.means_), _lib.ptr(g.covariances_), ctypes.byref(it),
                                                 ctypes.byref(lb), ctypes.byref(conv), ctypes.byref(kit)))
    else:
        _lib.check(ctx, _lib.lib.kwy_gmm_fit_comm_dev(ctx.handle, _lib.c_vp(Xd.data_ptr()), n, D, M, int(max_iter),
                                                      float(tol), float(reg_covar), int(random_state),
                                                      ctypes.cast(ctypes.pointer(kc), ctypes.c_void_p), _lib.ptr(g.weights_),
                                                      _lib.ptr(g.means_), _lib.ptr(g.covariances_), ctypes.byref(it),
                                                      ctypes.byref(lb), ctypes.byref(conv), ctypes.byref(kit)))
    del alive
    g.n_iter_, g.lower_bound_, g.converged_, g.kmeans_n_iter_ = it.value, lb.value, bool(conv.value), kit.value
    return g


class GaussianMixtureHIP:
    """Drop-in for the sklearn GaussianMixture object held by GMMFeatureConverter
    (full covariance, one initialisation)."""
    covariance_type = 'full'

    def __init__(self, n_components=1, covariance_type='full', tol=1e-3, reg_covar=1e-6, max_iter=100,
                 n_init=1, init_params='kmeans', random_state=None, verbose=0, device_index=0, sort_rows=True, **unused):
        if covariance_type != 'full' or n_init != 1 or init_params != 'kmeans':
            raise NotImplementedError('GaussianMixtureHIP implements the configuration the reference uses: '
                                      "covariance_type='full', n_init=1, init_params='kmeans'")
        self.n_components, self.tol, self.reg_covar, self.max_iter = n_components, tol, reg_covar, max_iter
        self.random_state, self.verbose, self.device_index = random_state, verbose, device_index
        self.sort_rows = sort_rows        # group the rows by their k-means cluster before EM (HipStats.sort_rows_by_component)

    def fit(self, X, y=None, stats=None, labels=None):
        """X: the local shard, (n, D) numpy array or a float64 device tensor.  labels: optional initial hard
        assignments of the local rows (skips the k-means run)."""
        import torch
        if stats is None:
            stats = HipStats(X, self.n_components, device_index=self.device_index)
        comm = Comm()
        with stats.scope():
            n_total = comm.all_reduce(torch.tensor([float(stats.n)], dtype=torch.float64, device=stats.dev))
            n_total = float(n_total.item())
        if n_total < self.n_components:
            raise ValueError(f'Expected n_samples >= n_components but got n_components = {self.n_components}, '
                             f'n_samples = {int(n_total)}')
        if self.verbose:
            print('Initialization 0')
        if labels is not None:
            with stats.scope():
                stats.set_resp_from_labels(labels)
        else:
            self.kmeans_n_iter_, self.kmeans_centers_ = kmeans_init(stats, self.n_components, self.random_state,
                                                                    verbose=self.verbose > 1)
        if self.sort_rows and hasattr(stats, 'sort_rows_by_component'):
            with stats.scope():
                stats.sort_rows_by_component()
        self.lower_bound_, self.converged_, self.n_iter_ = em_fit(
            stats, n_total, self.max_iter, self.tol, self.reg_covar, self.verbose > 1)
        if self.verbose:
            print(f'Initialization converged: {self.converged_}')
        self.weights_, self.means_, self.covariances_ = stats.get_params()
        return self
