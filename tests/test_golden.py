"""Committed golden vectors (tests/golden/golden.npz, made by tests/golden/make_golden.py).

CPU half: the oracle still reproduces them (guards the checker against drift; tolerance 1e-9
relative because libm picks CPU-specific exp/log/cos kernels).  GPU half: the HIP kernels, through
the C ABI, reproduce them on a box that has neither the reference nor its third-party stack; same
tolerances as tests/test_world_gpu.py and tests/test_backends_gpu.py.
"""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden.npz'))
FS = int(G['fs'])


def close(a, b, rel=1e-9):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    assert np.abs(a - b).max() <= rel * max(np.abs(b).max(), 1e-300)


def spectrum_close(got, ref):
    assert np.abs(got - ref).max() / np.abs(ref).max() <= 1e-8
    live = ref >= ref.max(axis=1, keepdims=True) * 1e-10
    assert np.abs(np.log(got[live]) - np.log(ref[live])).max() <= 1e-3


# ----------------------------------------------------------------------------- CPU: the oracle
def test_oracle_world_vectors():
    from oracle import oracle as ko
    x, t, fr = G['x'], G['t'], G['frames']
    f0d, t2 = ko.dio(x, FS)
    close(t2, t, 0)
    close(f0d, G['f0_dio'])
    f0 = ko.stonemask(x, f0d, t, FS)
    close(f0, G['f0'])
    sp, ap = ko.cheaptrick(x, G['f0'], t, FS), ko.d4c(x, G['f0'], t, FS)
    spectrum_close(sp[fr], G['sp_rows'])
    close(sp.sum(axis=1), G['sp_sum'], 1e-8)
    assert np.abs(ap[fr] - G['ap_rows']).max() <= 1e-7
    assert np.abs(ap.mean(axis=1) - G['ap_mean']).max() <= 1e-7
    y = ko.synthesize(G['f0'], sp, ap, FS, 5.0)
    assert np.sqrt(np.mean((y - G['y']) ** 2)) <= 1e-9


def test_oracle_mcep_dtw_mlpg_vectors():
    from oracle import oracle as ko
    assert ko.mcepalpha(FS) == float(G['alpha'])
    x, t = G['x'], G['t']
    sp = ko.cheaptrick(x, G['f0'], t, FS)
    mc = ko.sp2mc(sp, 24, float(G['alpha']))
    assert np.abs(mc - G['mc']).max() <= 1e-7
    close(ko.mc2sp(G['mc'][G['frames']], float(G['alpha']), 2 * (sp.shape[1] - 1)), G['sp_back'], 1e-9)
    dist, path = ko.fastdtw(G['feat_x'], G['feat_y'], radius=4, dist=2)
    assert np.array_equal(np.asarray(path, dtype=np.int32), G['dtw_path'])
    assert abs(dist - float(G['dtw_dist'])) <= 1e-9
    for diff, key in ((False, 'conv'), (True, 'conv_diff')):
        got = ko.gmm_mlpg(G['conv_in'], G['gmm_weights'], G['gmm_means'], G['gmm_covs'], diff=diff)
        assert np.abs(got - G[key]).max() <= 1e-9


def test_oracle_mlsa_codec_vectors():
    from oracle import oracle as ko
    hop = int(G['mlsa_hop'])
    b = ko.mc2b(G['mlsa_mc'], float(G['alpha']))
    assert np.abs(b - G['mlsa_b']).max() <= 1e-15
    y = ko.mlsa_synthesis(np.ascontiguousarray(G['x'][:len(G['mlsa_y'])]), G['mlsa_b'], float(G['alpha']), hop)
    assert np.abs(y - G['mlsa_y']).max() <= 1e-12 * np.abs(G['mlsa_y']).max()
    coded = ko.code_aperiodicity(np.ascontiguousarray(G['ap_rows']), FS)
    assert np.abs(coded - G['ap_coded']).max() <= 1e-9
    assert np.abs(ko.decode_aperiodicity(G['ap_coded'], FS, 512) - G['ap_decoded']).max() <= 1e-12


# ----------------------------------------------------------------------------- GPU: libkwy.so
@pytest.mark.gpu
def test_hip_world_vectors():
    from kwiiyatta_amd.backend import world as kw
    x, t, fr = G['x'], G['t'], G['frames']
    f0d, t2 = kw.dio(x, FS)
    assert np.array_equal(t2, t)
    assert np.abs(f0d - G['f0_dio']).max() <= 1e-6          # Hz
    f0 = kw.stonemask(x, G['f0_dio'], t, FS)
    assert np.abs(f0 - G['f0']).max() <= 1e-6
    sp, ap = kw.cheaptrick(x, G['f0'], t, FS), kw.d4c(x, G['f0'], t, FS)
    spectrum_close(sp[fr], G['sp_rows'])
    assert np.abs(sp.sum(axis=1) - G['sp_sum']).max() <= 1e-8 * G['sp_sum'].max()
    assert np.abs(ap[fr] - G['ap_rows']).max() <= 1e-4
    assert np.abs(ap.mean(axis=1) - G['ap_mean']).max() <= 1e-5
    y = kw.synthesize(G['f0'], sp, ap, FS, 5.0)
    assert y.shape == G['y'].shape
    assert np.sqrt(np.mean((y - G['y']) ** 2)) <= 1e-4      # the north-star criterion


@pytest.mark.gpu
def test_hip_mcep_dtw_mlpg_vectors():
    from kwiiyatta_amd.backend import dtw, mlpg, sptk, world as kw

    class Gmm:
        weights_, means_, covariances_ = G['gmm_weights'], G['gmm_means'], G['gmm_covs']
        covariance_type = 'full'

    sp = kw.cheaptrick(G['x'], G['f0'], G['t'], FS)
    assert np.abs(sptk.sp2mc(sp, 24, float(G['alpha'])) - G['mc']).max() <= 1e-6
    back = sptk.mc2sp(np.ascontiguousarray(G['mc'][G['frames']]), float(G['alpha']), 2 * (sp.shape[1] - 1))
    close(back, G['sp_back'], 1e-9)
    dist, path = dtw.fastdtw(G['feat_x'], G['feat_y'], radius=4, dist=2)
    assert np.array_equal(np.asarray(path, dtype=np.int32), G['dtw_path'])      # index work: exact
    assert abs(dist - float(G['dtw_dist'])) <= 1e-9
    X = mlpg.delta_features(G['conv_in'], mlpg.DELTA_WINDOWS)
    for diff, key in ((False, 'conv'), (True, 'conv_diff')):
        got = mlpg.MLPG(Gmm, windows=mlpg.DELTA_WINDOWS, diff=diff).transform(X)
        assert np.abs(got - G[key]).max() <= 1e-9


@pytest.mark.gpu
def test_hip_mlsa_codec_vectors():
    from kwiiyatta_amd.backend import sptk, world as kw
    hop, alpha = int(G['mlsa_hop']), float(G['alpha'])
    assert np.abs(sptk.mc2b(G['mlsa_mc'], alpha) - G['mlsa_b']).max() <= 1e-15
    x = np.ascontiguousarray(G['x'][:len(G['mlsa_y'])])
    y = sptk.Synthesizer(sptk.MLSADF(order=24, alpha=alpha), hopsize=hop).synthesis(x, G['mlsa_b'])
    assert np.abs(y - G['mlsa_y']).max() <= 1e-11 * np.abs(G['mlsa_y']).max()
    assert np.abs(kw.code_aperiodicity(np.ascontiguousarray(G['ap_rows']), FS) - G['ap_coded']).max() <= 1e-9
    assert np.abs(kw.decode_aperiodicity(G['ap_coded'], FS, 512) - G['ap_decoded']).max() <= 1e-12
