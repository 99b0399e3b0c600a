import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DATA = os.path.join(ROOT, 'tests', 'data')
CLB_DIR = os.path.join(DATA, 'cmu_us_clb_arctic', 'wav')
SLT_DIR = os.path.join(DATA, 'cmu_us_slt_arctic', 'wav')
CLB_WAV = os.path.join(CLB_DIR, 'arctic_a0001.wav')
CLB_WAV2 = os.path.join(CLB_DIR, 'arctic_a0002.wav')
SLT_WAV = os.path.join(SLT_DIR, 'arctic_a0001.wav')


def clb_variant(suffix):
    return os.path.join(DATA, f'cmu_us_clb_arctic.{suffix}', 'wav', 'arctic_a0001.wav')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def round_equal(expect, actual, sig_dig=2):
    """The reference's check.round_equal (tests/conftest.py:46-51): `actual`
    truncated to `sig_dig` significant digits equals `expect`."""
    import math
    eps = math.pow(10, math.floor(math.log10(abs(expect)) - sig_dig + 1)) if expect != 0 else 0
    return expect <= actual < expect + eps


@pytest.fixture(scope='session')
def gpu_ctx():
    from kwiiyatta_amd import _lib
    return _lib.default_context()


def _install_oracle_backend(monkeypatch):
    """Route the pyworld/pysptk/fastdtw/nnmnkwii-shaped backend modules to the CPU
    oracle, so the HOST logic of the package can be exercised without a GPU.
    Test-only: the product never does this."""
    import numpy as np
    from oracle import oracle as ko
    from kwiiyatta_amd.backend import dtw, mlpg, resample, sptk, world

    def cheaptrick(x, f0, t, fs, q1=-0.15, f0_floor=71.0, fft_size=None, ctx=None, out_div=1.0):
        sp = ko.cheaptrick(x, f0, t, fs, q1=q1, f0_floor=f0_floor, fft_size=fft_size)
        if out_div != 1.0:
            sp /= out_div
        return sp

    def synthesize(f0, sp, ap, fs, frame_period=5.0, ctx=None, sp_mul=1.0):
        return ko.synthesize(f0, np.ascontiguousarray(sp * sp_mul) if sp_mul != 1.0 else sp, ap, fs,
                             frame_period)

    class OracleMLPG:
        def __init__(self, gmm, windows=None, swap=False, diff=False, ctx=None):
            self.gmm, self.diff = gmm, diff
            self.static_dim = gmm.means_.shape[1] // 6

        def transform(self, src):
            d = self.static_dim
            return ko.gmm_mlpg(np.ascontiguousarray(src[:, :d]), self.gmm.weights_, self.gmm.means_,
                               self.gmm.covariances_, diff=self.diff)

    monkeypatch.setattr(world, 'dio', lambda x, fs, ctx=None, **kw: ko.dio(x, fs, **kw))
    monkeypatch.setattr(world, 'stonemask', lambda x, f0, t, fs, ctx=None: ko.stonemask(x, f0, t, fs))
    monkeypatch.setattr(world, 'cheaptrick', cheaptrick)
    monkeypatch.setattr(world, 'd4c', lambda x, f0, t, fs, ctx=None, **kw: ko.d4c(x, f0, t, fs, **kw))
    monkeypatch.setattr(world, 'synthesize', synthesize)
    monkeypatch.setattr(sptk, 'sp2mc', lambda sp, order, alpha, ctx=None: ko.sp2mc(sp, order, alpha))
    monkeypatch.setattr(sptk, 'mc2sp', lambda mc, alpha, fftlen, ctx=None: ko.mc2sp(mc, alpha, fftlen))
    monkeypatch.setattr(world, 'code_aperiodicity', lambda ap, fs, ctx=None: ko.code_aperiodicity(ap, fs))
    monkeypatch.setattr(world, 'decode_aperiodicity',
                        lambda c, fs, fft_size, ctx=None: ko.decode_aperiodicity(c, fs, fft_size))
    monkeypatch.setattr(sptk, 'mc2b', lambda mc, alpha=0.35, ctx=None: ko.mc2b(mc, alpha))
    monkeypatch.setattr(sptk.Synthesizer, 'synthesis',
                        lambda self, source, b: ko.mlsa_synthesis(source, b, self.filt.alpha, self.hopsize,
                                                                  self.filt.pd))
    monkeypatch.setattr(resample, 'stretch_log', lambda rows, new_bins, ctx=None: ko.stretch_log(rows, new_bins))
    monkeypatch.setattr(dtw, 'fastdtw', lambda x, y, radius=1, dist=2, ctx=None: ko.fastdtw(x, y, radius, dist))
    monkeypatch.setattr(mlpg, 'MLPG', OracleMLPG)
    from kwiiyatta_amd.converter import gmm as gmm_mod
    monkeypatch.setattr(gmm_mod, 'MLPG', OracleMLPG)
    import sklearn.mixture
    monkeypatch.setattr(gmm_mod, 'GaussianMixture', sklearn.mixture.GaussianMixture)  # the reference's own fit


@pytest.fixture(params=['oracle', pytest.param('hip', marks=pytest.mark.gpu)])
def kwiiyatta(request, monkeypatch):
    """The host package, with its numerics served either by the CPU oracle
    (host-logic tests, no GPU) or by the HIP kernels (-m gpu)."""
    import kwiiyatta_amd
    if request.param == 'oracle':
        _install_oracle_backend(monkeypatch)
    return kwiiyatta_amd
