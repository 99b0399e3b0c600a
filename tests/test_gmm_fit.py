"""Converter fit (EM): the distributed driver on CPU (2 ranks, gloo) with the
test-only numpy statistics, against scikit-learn; and, under -m gpu, the HIP
kernels against the same numpy statistics and against scikit-learn's fit."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from numpy_em_stats import NumpyStats


def make_data(n=3000, D=12, M=4, seed=0):
    rng = np.random.default_rng(seed)
    centres = rng.standard_normal((M, D)) * 3
    A = rng.standard_normal((M, D, D)) * 0.4 + np.eye(D)
    lab = rng.integers(0, M, n)
    X = centres[lab] + np.einsum('nij,nj->ni', A[lab], rng.standard_normal((n, D)))
    return np.ascontiguousarray(X)


def sklearn_fit(X, M, seed=0, max_iter=100, tol=1e-3):
    from sklearn.mixture import GaussianMixture
    return GaussianMixture(n_components=M, covariance_type='full', max_iter=max_iter, tol=tol,
                           random_state=seed).fit(X)


def kmeans_labels(X, M, seed):
    from sklearn.cluster import KMeans
    from sklearn.utils import check_random_state
    km = KMeans(n_clusters=M, n_init=1, random_state=check_random_state(seed)).fit(X)
    return km.labels_, km.cluster_centers_, km.n_iter_


@pytest.mark.parametrize('n,D,M,seed', [(3000, 12, 4, 0), (5000, 30, 16, 1), (2500, 20, 64, 2)])
def test_kmeans_driver_matches_sklearn(n, D, M, seed):
    """The k-means initialisation (k-means++ with scikit-learn's random draws, Lloyd) reproduces
    sklearn.cluster.KMeans(n_init=1) -- what GaussianMixture(init_params='kmeans') runs
    (kwiiyatta/converter/gmm.py:14-23) -- label for label."""
    from kwiiyatta_amd.converter.gmm_fit import kmeans_init
    X = make_data(n, D, min(M, 8), seed=seed)
    labels, centres, n_iter = kmeans_labels(X, M, seed)
    st = NumpyStats(X, M)
    it, c = kmeans_init(st, M, seed)
    assert it == n_iter
    assert np.array_equal(st.labels, labels)
    assert np.abs(c - centres).max() <= 1e-12 * np.abs(centres).max()
    assert np.array_equal(st.resp.argmax(1), labels) and (st.resp.sum(1) == 1).all()


def test_kmeans_empty_cluster_relocation():
    """More centres than distinct points in a region: Lloyd empties a cluster and it is re-seeded from the row
    farthest from its centre, as sklearn does."""
    from kwiiyatta_amd.converter import gmm_fit
    import torch
    rng = np.random.default_rng(0)
    X = np.vstack([rng.standard_normal((200, 3)) * 0.01, rng.standard_normal((200, 3)) * 0.01 + 5])
    st = NumpyStats(X, 3)
    st.km_begin(torch.from_numpy(X.mean(0)))
    centres = torch.from_numpy(np.array([[-2.5, -2.5, -2.5], [2.5, 2.5, 2.5], [50.0, 50, 50]]))
    st.km_assign(centres)
    s = st.km_sums()
    assert s[2, 0] == 0
    gmm_fit._relocate_empty_clusters(st, gmm_fit.Comm(), s, centres, (s[:, 0] == 0).nonzero().flatten(), 0)
    assert s[2, 0] == 1 and s[:, 0].sum() == 400
    far = ((st.Xc - centres.numpy()[st.labels]) ** 2).sum(1).argmax()
    assert np.array_equal(s[2, 1:].numpy(), st.Xc[far])


def test_fit_driver_matches_sklearn_single_process():
    """Initialisation and EM loop together: sklearn's n_iter, lower bound and parameters."""
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP
    X = make_data()
    ref = sklearn_fit(X, 4)
    g = GaussianMixtureHIP(n_components=4, random_state=0).fit(X, stats=NumpyStats(X, 4))
    assert g.n_iter_ == ref.n_iter_ and g.converged_ == ref.converged_
    assert abs(g.lower_bound_ - ref.lower_bound_) < 1e-9
    assert np.allclose(g.weights_, ref.weights_, rtol=1e-8, atol=1e-12)
    assert np.allclose(g.means_, ref.means_, rtol=1e-7, atol=1e-9)
    assert np.allclose(g.covariances_, ref.covariances_, rtol=1e-6, atol=1e-9)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


TWO_RANK = dict(n=4000, D=16, M=12, seed=5, max_iter=30)


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP
    from numpy_em_stats import NumpyStats as NS
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    c = TWO_RANK
    X = make_data(c['n'], c['D'], 6, seed=c['seed'])
    cut = [0, 1700, c['n']]                      # uneven contiguous shards: global row order = rank order
    Xl = np.ascontiguousarray(X[cut[rank]:cut[rank + 1]])
    st = NS(Xl, c['M'])
    g = GaussianMixtureHIP(n_components=c['M'], random_state=c['seed'], max_iter=c['max_iter'])
    g.fit(Xl, stats=st)
    q.put((rank, g.n_iter_, g.lower_bound_, g.weights_, g.means_, g.covariances_, g.kmeans_n_iter_,
           st.resp.argmax(1) if False else None, g.kmeans_centers_))
    dist.barrier()
    dist.destroy_process_group()


def test_fit_two_ranks_gloo():
    """Rows sharded over 2 ranks: both ranks end with the same model, and it equals the single-process fit of
    the whole matrix INCLUDING the k-means initialisation (same seeding draws, same Lloyd iterations), and
    scikit-learn's own fit."""
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, it0, lb0, w0, m0, c0, kit0, _, kc0), (_, it1, lb1, w1, m1, c1, kit1, _, kc1) = res
    assert it0 == it1 and lb0 == lb1 and kit0 == kit1
    assert np.array_equal(w0, w1) and np.array_equal(m0, m1) and np.array_equal(c0, c1)
    assert np.array_equal(kc0, kc1)

    c = TWO_RANK
    X = make_data(c['n'], c['D'], 6, seed=c['seed'])
    one = GaussianMixtureHIP(n_components=c['M'], random_state=c['seed'], max_iter=c['max_iter'])
    one.fit(X, stats=NumpyStats(X, c['M']))
    assert one.kmeans_n_iter_ == kit0
    assert np.abs(one.kmeans_centers_ - kc0).max() <= 1e-12 * np.abs(kc0).max()
    assert one.n_iter_ == it0 and abs(one.lower_bound_ - lb0) < 1e-10
    assert np.allclose(one.weights_, w0, rtol=1e-10) and np.allclose(one.means_, m0, rtol=1e-9, atol=1e-12)
    assert np.allclose(one.covariances_, c0, rtol=1e-8, atol=1e-12)
    ref = sklearn_fit(X, c['M'], seed=c['seed'], max_iter=c['max_iter'])
    assert ref.n_iter_ == it0 and abs(ref.lower_bound_ - lb0) < 1e-9
    assert np.allclose(ref.means_, m0, rtol=1e-7, atol=1e-9)


def _hip_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    c = TWO_RANK_HIP
    X = make_data(c['n'], c['D'], 6, seed=c['seed'])
    cut = [0, c['cut'], c['n']]
    g = GaussianMixtureHIP(n_components=c['M'], random_state=c['seed'], max_iter=c['max_iter'], device_index=0)
    g.fit(np.ascontiguousarray(X[cut[rank]:cut[rank + 1]]))      # the product's statistics class: HipStats
    q.put((rank, g.n_iter_, g.lower_bound_, g.weights_, g.means_, g.covariances_, g.kmeans_n_iter_, g.kmeans_centers_))
    dist.barrier()
    dist.destroy_process_group()


TWO_RANK_HIP = dict(n=9000, D=144, M=16, seed=3, max_iter=12, cut=3700)


@pytest.mark.gpu
def test_hip_fit_two_ranks_share_one_gpu():
    """The N-rank fit with the PRODUCT's statistics class: two processes (gloo for the collectives, both on cuda:0)
    fit uneven shards with HipStats -- foreign (-1) rows in the seeding, the external-stream ordering around the
    collectives, the device-side all-gathers -- and end with the same model as one process on all rows, and as
    scikit-learn."""
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hip_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, it0, lb0, w0, m0, c0, kit0, kc0), (_, it1, lb1, w1, m1, c1, kit1, kc1) = res
    assert it0 == it1 and kit0 == kit1 and abs(lb0 - lb1) <= 1e-12 * abs(lb0)
    assert np.allclose(w0, w1, rtol=1e-12) and np.allclose(m0, m1, rtol=1e-12, atol=1e-14)
    c = TWO_RANK_HIP
    X = make_data(c['n'], c['D'], 6, seed=c['seed'])
    one = GaussianMixtureHIP(n_components=c['M'], random_state=c['seed'], max_iter=c['max_iter']).fit(X)
    assert one.kmeans_n_iter_ == kit0
    assert np.abs(one.kmeans_centers_ - kc0).max() <= 1e-11 * np.abs(kc0).max()
    assert one.n_iter_ == it0 and abs(one.lower_bound_ - lb0) <= 1e-9 * abs(lb0)
    assert np.allclose(one.weights_, w0, rtol=1e-9) and np.allclose(one.means_, m0, rtol=1e-8, atol=1e-11)
    assert np.allclose(one.covariances_, c0, rtol=1e-7, atol=1e-10)
    ref = sklearn_fit(X, c['M'], seed=c['seed'], max_iter=c['max_iter'])
    assert ref.n_iter_ == it0 and np.allclose(ref.means_, m0, rtol=1e-6, atol=1e-8)


def test_rccl_path_reduces_device_tensors_only():
    """Under backend 'nccl' a host tensor must never reach all_reduce (RCCL has no CPU backend): the
    communicator refuses it instead of letting torch.distributed raise half-way through a fit."""
    from kwiiyatta_amd.converter import gmm_fit
    import torch

    class FakeDist:
        class ReduceOp:
            SUM = 0

        def get_rank(self): return 0
        def get_world_size(self): return 2
        def get_backend(self): return 'nccl'
        def all_reduce(self, t, op=None): raise AssertionError('host tensor reached RCCL')

    c = gmm_fit.Comm.__new__(gmm_fit.Comm)
    c.dist, c.rank, c.world, c.device_native = FakeDist(), 0, 2, True
    with pytest.raises(RuntimeError):
        c.all_reduce(torch.zeros(3))


@pytest.mark.gpu
@pytest.mark.parametrize('n,D,M', [(3000, 12, 4), (5000, 144, 8), (2000, 30, 64), (2500, 150, 5), (1500, 160, 3),
                                   (100, 20, 3), (20000, 144, 64), (777, 97, 7)])
def test_hip_statistics_match_numpy(n, D, M):
    """E-step, sums, covariance statistics and finalisation of the HIP kernels vs numpy."""
    from kwiiyatta_amd.converter.gmm_fit import HipStats
    X = make_data(n, D, min(M, 8), seed=1)
    rng = np.random.default_rng(2)
    labels = rng.integers(0, M, n)
    labels[:M] = np.arange(M)
    hs, ns = HipStats(X, M), NumpyStats(X, M)
    with hs.scope():
        for s in (hs, ns):
            s.set_resp_from_labels(labels)
        sh, sn = hs.sums().cpu().numpy(), ns.sums().numpy()
        assert np.allclose(sh, sn, rtol=1e-12, atol=1e-12)
        hs.means_from(hs.stats)
        ns.means_from(ns.sums())
        ch, cn = hs.cov().cpu().numpy(), ns.cov().numpy()
        assert np.allclose(ch, cn, rtol=1e-10, atol=1e-10)
        hs.finalize(hs.stats, hs.sxx, 1e-3)
        import torch
        ns.finalize(torch.from_numpy(sn), torch.from_numpy(cn), 1e-3)
        wh, mh, covh = hs.get_params()
        assert np.allclose(wh, ns.weights, rtol=1e-12) and np.allclose(mh, ns.means, rtol=1e-10, atol=1e-12)
        assert np.allclose(covh, ns.covs, rtol=1e-9, atol=1e-10)
        llh, lln = float(hs.estep().item()), float(ns.estep().item())
        assert abs(llh - lln) <= 1e-9 * abs(lln)
        assert np.allclose(hs.resp.cpu().numpy(), ns.resp, rtol=1e-7, atol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize('n,D,M,seed', [(3000, 12, 4, 0), (20000, 144, 64, 1), (5000, 97, 130, 2), (300, 8, 3, 3)])
def test_hip_kmeans_matches_sklearn(n, D, M, seed):
    """kwy_km_* through the driver against sklearn.cluster.KMeans(n_init=1): same seeds, same labels."""
    from kwiiyatta_amd.converter.gmm_fit import HipStats, kmeans_init
    X = make_data(n, D, min(M, 8), seed=seed)
    labels, centres, n_iter = kmeans_labels(X, M, seed)
    st = HipStats(X, M)
    it, c = kmeans_init(st, M, seed)
    with st.scope():
        got = st.labels.cpu().numpy()
        resp = st.resp.cpu().numpy()
    assert it == n_iter
    assert np.array_equal(got, labels)
    assert np.abs(c - centres).max() <= 1e-11 * np.abs(centres).max()
    assert np.array_equal(resp.argmax(1), labels) and (resp.sum(1) == 1).all() and (resp.max(1) == 1).all()


@pytest.mark.gpu
@pytest.mark.parametrize('batch', [1, 3, 16])
def test_hip_lloyd_on_the_device_equals_the_host_loop(batch, monkeypatch):
    """kwy_km_lloyd_dev (the stopping decision taken on the device, `batch` iterations per read-back) against the loop
    with one host round trip per iteration: same iteration count, labels and centres -- on blobs, and on the data
    whose k-means++ seeding leaves clusters empty (the iteration is handed back to the host for the relocation)."""
    from kwiiyatta_amd.converter import gmm_fit
    from kwiiyatta_amd.converter.gmm_fit import HipStats, kmeans_init
    rng = np.random.default_rng(4)
    cases = [(make_data(6000, 24, 8, seed=5), 12, 2)]
    for seed in range(4):
        X = np.repeat(rng.standard_normal((10, 6)) * 20, 40, axis=0) + rng.standard_normal((400, 6)) * 1e-3
        cases.append((X, 16, seed))
    for X, M, seed in cases:
        monkeypatch.setattr(gmm_fit, 'LLOYD_BATCH', batch)
        dev = HipStats(X, M)
        it_d, c_d = kmeans_init(dev, M, seed)
        with dev.scope():
            lab_d = dev.labels.cpu().numpy()
        host = HipStats(X, M)
        monkeypatch.delattr(HipStats, 'km_lloyd')                 # hasattr fails -> the per-iteration loop
        it_h, c_h = kmeans_init(host, M, seed)
        monkeypatch.undo()
        with host.scope():
            lab_h = host.labels.cpu().numpy()
        assert it_d == it_h
        assert np.array_equal(lab_d, lab_h)
        assert np.array_equal(c_d, c_h)


@pytest.mark.gpu
def test_hip_kmeans_blocks_match_numpy():
    """the individual k-means blocks against their numpy restatement (shard-local parts: colstats, centring,
    candidate distances, the cumulative-sum search incl. the not-mine / clipped cases, assignment, update)"""
    import torch
    from kwiiyatta_amd.converter.gmm_fit import HipStats
    n, D, M = 7001, 144, 64
    X = make_data(n, D, 8, seed=4)
    hs, ns = HipStats(X, M), NumpyStats(X, M)
    with hs.scope():
        s_h, s_n = hs.km_colstats(None).cpu().numpy(), ns.km_colstats(None).numpy()
        assert np.allclose(s_h, s_n, rtol=1e-12, atol=1e-9)
        mean = X.mean(0)
        hs.km_begin(torch.from_numpy(mean).cuda())
        ns.km_begin(torch.from_numpy(mean))
        assert np.allclose(hs.Xc.cpu().numpy(), ns.Xc, rtol=0, atol=0)
        assert np.allclose(hs.xsq.cpu().numpy(), ns.xsq, rtol=1e-13)
        cand = ns.Xc[[5, 77, 3000, 6999, 12, 4000]]
        p_h = hs.km_candidates(torch.from_numpy(cand).cuda(), use_closest=False).cpu().numpy()
        p_n = ns.km_candidates(torch.from_numpy(cand), use_closest=False).numpy()
        assert np.allclose(p_h, p_n, rtol=1e-12)
        assert np.allclose(hs.newd[:6].cpu().numpy(), ns.newd, rtol=1e-11, atol=1e-9)
        best = torch.tensor([2])
        hs.km_accept(best.cuda())
        ns.km_accept(best)
        assert np.array_equal(hs.closest.cpu().numpy(), hs.newd[2].cpu().numpy())
        tot_h = float(hs.km_closest_total().item())
        assert abs(tot_h - ns.closest.sum()) <= 1e-12 * tot_h
        cum = np.cumsum(ns.closest)
        vals = np.array([0.0, cum[0] * 0.5, cum[10] * (1 - 1e-9), cum[3333] * (1 + 1e-12), cum[-1] * 0.999999, cum[-1] * 1.01])
        for lo, first, last in ((0.0, True, True), (0.0, True, False), (cum[-1] * 0.25, False, False), (cum[-1] * 0.25, False, True)):
            hi = lo + tot_h
            t64 = lambda v: torch.tensor([v], dtype=torch.float64)     # noqa: E731
            want = ns.km_pick(t64(lo), t64(hi), torch.from_numpy(vals), first, last).numpy()
            got = hs.km_pick(t64(lo).cuda(), t64(hi).cuda(), torch.from_numpy(vals).cuda(), first, last).cpu().numpy()
            assert np.array_equal(got, want), (lo, first, last, got, want)
        # the edges of the search (the k-means++ seeding once aborted in a torch gather on an out-of-range row,
        # DESIGN.md section 6): a draw at or beyond the total potential -> the LAST row, never row n; a value owned
        # by another shard -> -1, which the driver must not use as an index; a value exactly on the upper bound
        # belongs to this shard and not to the next one
        lo, hi = 3.0 * tot_h, 4.0 * tot_h
        edge = np.array([hi, hi * (1 + 1e-15) + 1e-300, lo, lo * (1 - 1e-15), 10 * hi, np.nextafter(lo, np.inf)])
        t = lambda v: torch.tensor([v], dtype=torch.float64).cuda()     # noqa: E731
        mid = hs.km_pick(t(lo), t(hi), torch.from_numpy(edge).cuda(), False, False).cpu().numpy()
        assert mid[0] == n - 1 and mid[1] == -1 and mid[2] == -1 and mid[3] == -1 and mid[4] == -1 and mid[5] == 0
        nxt = hs.km_pick(t(hi), t(hi + tot_h), torch.from_numpy(edge).cuda(), False, True).cpu().numpy()
        assert nxt[0] == -1 and nxt[1] == 0 and nxt[4] == n - 1          # the boundary value has exactly one owner
        assert ((mid >= -1) & (mid < n)).all() and ((nxt >= -1) & (nxt < n)).all()
        p_h = hs.km_candidates(torch.from_numpy(cand[:3].copy()).cuda(), use_closest=True).cpu().numpy()
        p_n = ns.km_candidates(torch.from_numpy(cand[:3].copy()), use_closest=True).numpy()
        assert np.allclose(p_h, p_n, rtol=1e-12)
        centres = ns.Xc[np.random.default_rng(0).choice(n, M, replace=False)].copy()
        ch_h = int(hs.km_assign(torch.from_numpy(centres).cuda()).item())
        ch_n = int(ns.km_assign(torch.from_numpy(centres)).item())
        assert ch_h == ch_n == n
        assert np.array_equal(hs.labels.cpu().numpy(), ns.labels)
        st_h, st_n = hs.km_sums(), ns.km_sums()
        assert np.allclose(st_h.cpu().numpy(), st_n.numpy(), rtol=1e-12, atol=1e-12)
        new_h, new_n = torch.empty((M, D), dtype=torch.float64, device='cuda'), torch.empty((M, D), dtype=torch.float64)
        sh_h = hs.km_update(st_h, torch.from_numpy(centres).cuda(), new_h).cpu().numpy()
        sh_n = ns.km_update(st_n, torch.from_numpy(centres), new_n).numpy()
        assert np.allclose(new_h.cpu().numpy(), new_n.numpy(), rtol=1e-12, atol=1e-14) and np.allclose(sh_h, sh_n, rtol=1e-10)
        assert int(hs.km_assign(torch.from_numpy(centres).cuda()).item()) == 0      # same centres: nothing changes


@pytest.mark.gpu
def test_hip_fit_matches_sklearn_bench_shape():
    """D = 144, M = 64 (the reference's converter: 2 x 3 x 24 joint dims, 64 components), initialisation
    included, against sklearn.mixture.GaussianMixture.fit."""
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP
    X = make_data(20000, 144, 8, seed=6)
    ref = sklearn_fit(X, 64, seed=1, max_iter=8)
    g = GaussianMixtureHIP(n_components=64, random_state=1, max_iter=8).fit(X)
    assert g.n_iter_ == ref.n_iter_
    assert abs(g.lower_bound_ - ref.lower_bound_) < 1e-8 * abs(ref.lower_bound_)
    assert np.allclose(g.weights_, ref.weights_, rtol=1e-6, atol=1e-10)
    assert np.allclose(g.means_, ref.means_, rtol=1e-6, atol=1e-8)
    assert np.allclose(g.covariances_, ref.covariances_, rtol=1e-5, atol=1e-8)


@pytest.mark.gpu
def test_hip_fit_matches_sklearn():
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP
    X = make_data(6000, 24, 8, seed=3)
    ref = sklearn_fit(X, 8)
    g = GaussianMixtureHIP(n_components=8, random_state=0).fit(X)
    assert g.n_iter_ == ref.n_iter_ and g.converged_ == ref.converged_
    assert abs(g.lower_bound_ - ref.lower_bound_) < 1e-8
    assert np.allclose(g.weights_, ref.weights_, rtol=1e-6, atol=1e-10)
    assert np.allclose(g.means_, ref.means_, rtol=1e-6, atol=1e-8)
    assert np.allclose(g.covariances_, ref.covariances_, rtol=1e-5, atol=1e-8)


@pytest.mark.gpu
def test_hip_fit_rejects_degenerate_covariance():
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP
    X = np.zeros((100, 4))
    X[:, 0] = np.arange(100)
    with pytest.raises(ValueError):
        GaussianMixtureHIP(n_components=2, random_state=0, reg_covar=0.0).fit(X)


@pytest.mark.gpu
@pytest.mark.parametrize('n,D,M,seed', [(6000, 24, 8, 0), (20000, 144, 64, 1), (900, 12, 3, 7)])
def test_single_call_fit_matches_driver_and_sklearn(n, D, M, seed):
    """kwy_gmm_fit_dev -- initialisation and EM as ONE C call (numpy's RandomState draws reproduced in the library) --
    against the Python driver over the same kernels and against scikit-learn: same iteration counts, same model."""
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP, fit_one_call
    X = make_data(n, D, min(M, 8), seed=seed + 10)
    one = fit_one_call(X, M, max_iter=10, random_state=seed)
    drv = GaussianMixtureHIP(n_components=M, random_state=seed, max_iter=10).fit(X)
    ref = sklearn_fit(X, M, seed=seed, max_iter=10)
    assert one.kmeans_n_iter_ == drv.kmeans_n_iter_
    assert one.n_iter_ == drv.n_iter_ == ref.n_iter_ and one.converged_ == drv.converged_
    assert abs(one.lower_bound_ - drv.lower_bound_) <= 1e-10 * abs(drv.lower_bound_)
    assert np.allclose(one.weights_, drv.weights_, rtol=1e-10) and np.allclose(one.means_, drv.means_, rtol=1e-9, atol=1e-12)
    assert np.allclose(one.covariances_, drv.covariances_, rtol=1e-8, atol=1e-12)
    assert np.allclose(one.means_, ref.means_, rtol=1e-6, atol=1e-8)
    assert np.allclose(one.covariances_, ref.covariances_, rtol=1e-5, atol=1e-8)


def _comm_worker(rank, world, port, q, backend):
    """one rank of the C-entry fit: kwy_gmm_fit_comm_dev with torch.distributed's group as the communicator"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP, fit_one_call
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    if backend == 'nccl':
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    c = TWO_RANK_HIP
    X = make_data(c['n'], c['D'], 6, seed=c['seed'])
    cut = [0, c['cut'], c['n']] if world == 2 else [0, c['n']]
    shard = np.ascontiguousarray(X[cut[rank]:cut[rank + 1]])
    g = fit_one_call(shard, c['M'], max_iter=c['max_iter'], random_state=c['seed'], distributed=True)
    # the Python driver (HipStats + Comm) under the same process group
    d = GaussianMixtureHIP(n_components=c['M'], random_state=c['seed'], max_iter=c['max_iter'], device_index=0).fit(shard)
    q.put((rank, g.n_iter_, g.lower_bound_, g.weights_, g.means_, g.covariances_, g.kmeans_n_iter_,
           d.n_iter_, d.lower_bound_, d.means_, d.kmeans_n_iter_))
    dist.barrier()
    dist.destroy_process_group()


def _run_comm_workers(world, backend):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _check_against_one_rank(res):
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP
    c = TWO_RANK_HIP
    X = make_data(c['n'], c['D'], 6, seed=c['seed'])
    one = GaussianMixtureHIP(n_components=c['M'], random_state=c['seed'], max_iter=c['max_iter']).fit(X)
    for (_, it, lb, w, m, cv, kit, dit, dlb, dm, dkit) in res:
        assert it == dit == one.n_iter_ and kit == dkit == one.kmeans_n_iter_
        assert abs(lb - one.lower_bound_) <= 1e-9 * abs(lb) and abs(dlb - one.lower_bound_) <= 1e-9 * abs(lb)
        assert np.allclose(w, one.weights_, rtol=1e-9) and np.allclose(m, one.means_, rtol=1e-8, atol=1e-11)
        assert np.allclose(cv, one.covariances_, rtol=1e-7, atol=1e-10)
        assert np.allclose(dm, one.means_, rtol=1e-8, atol=1e-11)
    return one


@pytest.mark.gpu
def test_c_entry_fit_two_ranks_share_one_gpu():
    """kwy_gmm_fit_comm_dev (the multi-rank fit as ONE C call, the communicator an all-reduce callback) run by two
    processes on cuda:0 over uneven shards (gloo carries the callback's reductions): both ranks return the same model,
    the model of the Python driver under the same group, of a one-rank fit of all rows, and of scikit-learn."""
    res = _run_comm_workers(2, 'gloo')
    (_, it0, lb0, w0, m0, c0, kit0, *_), (_, it1, lb1, w1, m1, c1, kit1, *_) = res
    assert it0 == it1 and kit0 == kit1 and lb0 == lb1
    assert np.array_equal(w0, w1) and np.array_equal(m0, m1) and np.array_equal(c0, c1)
    _check_against_one_rank(res)
    c = TWO_RANK_HIP
    ref = sklearn_fit(make_data(c['n'], c['D'], 6, seed=c['seed']), c['M'], seed=c['seed'], max_iter=c['max_iter'])
    assert ref.n_iter_ == it0 and np.allclose(ref.means_, m0, rtol=1e-6, atol=1e-8)


@pytest.mark.gpu
def test_fit_under_rccl_world_size_one():
    """RCCL itself under the fit: a world-size-1 'nccl' process group on cuda:0 (the one GPU of the test box).  The
    Python driver (HipStats + Comm: device tensors, the fit's own stream current around the collectives) and the C entry
    (kwy_gmm_fit_comm_dev, whose callback ends in ncclAllReduce on the library's stream) both equal the fit without a
    process group.  Runs in a child process: the group must exist before the first collective and die with it."""
    res = _run_comm_workers(1, 'nccl')
    _check_against_one_rank(res)


@pytest.mark.gpu
def test_c_entry_relocates_two_empty_clusters_like_the_driver():
    """two clusters emptied in the same Lloyd iteration: the C entry and the Python driver pair the farthest rows with
    the empty clusters in the same order (largest distance first; scikit-learn's argpartition leaves that order
    unspecified, so its labels may differ there -- documented in kwy_fit_driver.hip)"""
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP, fit_one_call
    rng = np.random.default_rng(4)
    # 30 tight far-apart blobs of 3 points and 8 centres: k-means++ on so few distinct regions leaves several
    # clusters empty along the way for some seeds
    hits = 0
    for seed in range(12):
        X = np.repeat(rng.standard_normal((10, 6)) * 20, 40, axis=0) + rng.standard_normal((400, 6)) * 1e-3
        one = fit_one_call(X, 16, max_iter=3, random_state=seed)
        drv = GaussianMixtureHIP(n_components=16, random_state=seed, max_iter=3, reg_covar=1e-6).fit(X)
        assert one.kmeans_n_iter_ == drv.kmeans_n_iter_ and one.n_iter_ == drv.n_iter_
        assert np.allclose(one.means_, drv.means_, rtol=1e-9, atol=1e-9)
        hits += 1
    assert hits == 12
