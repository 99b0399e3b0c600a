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


def test_em_driver_matches_sklearn_single_process():
    """Same initial labels, same loop => sklearn's n_iter, lower bound and parameters."""
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


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP
    from kwiiyatta_amd.parallel import shard_indices
    from numpy_em_stats import NumpyStats as NS
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    X = make_data()
    mine = shard_indices(len(X), rank, world)
    Xl = np.ascontiguousarray(X[mine])
    g = GaussianMixtureHIP(n_components=4, random_state=0, max_iter=30)
    g.fit(Xl, stats=NS(Xl, 4))
    q.put((rank, g.n_iter_, g.lower_bound_, g.weights_, g.means_, g.covariances_))
    dist.barrier()
    dist.destroy_process_group()


def test_em_two_ranks_gloo():
    """Frames sharded over 2 ranks, statistics all-reduced: both ranks end with the
    same model, and it equals the single-process run from the same initial labels."""
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
    (_, it0, lb0, w0, m0, c0), (_, it1, lb1, w1, m1, c1) = res
    assert it0 == it1 and lb0 == lb1
    assert np.array_equal(w0, w1) and np.array_equal(m0, m1) and np.array_equal(c0, c1)

    # single-process reference with the labels the distributed initialisation produces
    from sklearn.cluster import KMeans
    from sklearn.utils import check_random_state
    from kwiiyatta_amd.parallel import shard_indices
    X = make_data()
    X0 = X[shard_indices(len(X), 0, world)]
    centres = KMeans(n_clusters=4, n_init=1, random_state=check_random_state(0)).fit(X0).cluster_centers_
    labels = ((X ** 2).sum(1)[:, None] - 2 * X @ centres.T + (centres ** 2).sum(1)[None, :]).argmin(1)

    class Fixed(GaussianMixtureHIP):
        def _initial_labels(self, X):
            return labels
    ref = Fixed(n_components=4, random_state=0, max_iter=30).fit(X, stats=NumpyStats(X, 4))
    assert ref.n_iter_ == it0
    assert abs(ref.lower_bound_ - lb0) < 1e-10
    assert np.allclose(ref.weights_, w0, rtol=1e-10) and np.allclose(ref.means_, m0, rtol=1e-9, atol=1e-12)
    assert np.allclose(ref.covariances_, c0, rtol=1e-8, atol=1e-12)


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
    for s in (hs, ns):
        s.set_resp_from_labels(labels)
    sh, sn = hs.sums().cpu().numpy(), ns.sums()
    assert np.allclose(sh, sn, rtol=1e-12, atol=1e-12)
    hs.means_from(hs.stats)
    ns.means_from(sn)
    ch, cn = hs.cov().cpu().numpy(), ns.cov()
    assert np.allclose(ch, cn, rtol=1e-10, atol=1e-10)
    hs.finalize(hs.stats, hs.sxx, 1e-3)
    ns.finalize(sn, cn, 1e-3)
    wh, mh, covh = hs.get_params()
    assert np.allclose(wh, ns.weights, rtol=1e-12) and np.allclose(mh, ns.means, rtol=1e-10, atol=1e-12)
    assert np.allclose(covh, ns.covs, rtol=1e-9, atol=1e-10)
    llh, lln = hs.estep(), ns.estep()
    assert abs(llh - lln) <= 1e-9 * abs(lln)
    assert np.allclose(hs.resp.cpu().numpy(), ns.resp, rtol=1e-7, atol=1e-10)


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
