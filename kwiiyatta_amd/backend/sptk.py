"""pysptk-shaped front-end of the HIP mel-cepstrum and MLSA kernels
(reference call sites /root/reference/kwiiyatta/vocoder/mcep.py:26,65,71 and
/root/reference/kwiiyatta/filter/mlsa.py:24-29)."""
import functools

import numpy as np

from .. import _lib
from .._lib import lib, ptr


def _rows(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.reshape(-1, a.shape[-1]), a.shape[:-1]


def sp2mc(powerspec, order, alpha, ctx=None):
    """power spectrum (..., K) -> mel-cepstrum (..., order+1), row-wise."""
    sp, lead = _rows(powerspec)
    ctx = ctx or _lib.default_context()
    mc = np.empty((sp.shape[0], order + 1))
    _lib.check(ctx, lib.kwy_sp2mc(ctx.handle, ptr(sp), sp.shape[0], sp.shape[1], int(order),
                                  float(alpha), ptr(mc)))
    return mc.reshape(lead + (order + 1,))


def mc2sp(mc, alpha, fftlen, ctx=None):
    """mel-cepstrum (..., order+1) -> power spectrum (..., fftlen/2+1), row-wise."""
    m, lead = _rows(mc)
    ctx = ctx or _lib.default_context()
    sp = np.empty((m.shape[0], fftlen // 2 + 1))
    _lib.check(ctx, lib.kwy_mc2sp(ctx.handle, ptr(m), m.shape[0], m.shape[1] - 1, float(alpha),
                                  int(fftlen), ptr(sp)))
    return sp.reshape(lead + (fftlen // 2 + 1,))


@functools.lru_cache(maxsize=None)
def mcepalpha(fs, start=0.0, stop=1.0, step=0.001, num_points=1000):
    """pysptk.util.mcepalpha: the all-pass constant whose warping is closest
    (RMS) to the mel scale at sampling rate fs.  Host-side grid search."""
    alphas = np.arange(start, stop, step)
    mel = np.log(1 + (fs / 2.0) / num_points * np.arange(num_points) / 1000.0) * (1000.0 / np.log(2))
    mel = mel / mel[-1]
    omega = np.pi / num_points * np.arange(num_points)
    a = alphas[:, None]
    with np.errstate(divide='ignore', invalid='ignore'):
        warp = np.arctan((1 - a * a) * np.sin(omega) / ((1 + a * a) * np.cos(omega) - 2 * a))
    warp[warp < 0] += np.pi
    warp = warp / warp[:, -1:]
    dist = np.sqrt(np.mean((mel[None, :] - warp) ** 2, axis=1))
    return float(alphas[np.argmin(dist)])


def mc2b(mc, alpha=0.35, ctx=None):
    """pysptk.mc2b: mel-cepstrum -> MLSA filter coefficients, row-wise."""
    m, lead = _rows(mc)
    ctx = ctx or _lib.default_context()
    b = np.empty_like(m)
    _lib.check(ctx, lib.kwy_mc2b(ctx.handle, ptr(m), m.shape[0], m.shape[1] - 1, float(alpha), ptr(b)))
    return b.reshape(lead + (m.shape[1],))


class MLSADF:
    """pysptk.synthesis.MLSADF: holds the filter description (the state lives on the GPU for the
    duration of one Synthesizer.synthesis call)."""

    def __init__(self, order=25, alpha=0.35, pd=4):
        if pd not in (4, 5):
            raise ValueError('4 or 5 pade approximations are supported')
        self.order, self.alpha, self.pd = int(order), float(alpha), int(pd)


class Synthesizer:
    """pysptk.synthesis.Synthesizer for an MLSADF: synthesis(source, b) filters `source` with the
    frame-wise coefficients b (T, order+1), interpolating inside each hop."""

    def __init__(self, filt, hopsize, ctx=None):
        if not isinstance(filt, MLSADF):
            raise NotImplementedError('only MLSADF, the filter the reference uses, is implemented')
        self.filt, self.hopsize, self.ctx = filt, int(hopsize), ctx

    def synthesis(self, source, b):
        x = _lib.as_f64(np.asarray(source, dtype=np.float64))
        b = np.ascontiguousarray(b, dtype=np.float64)
        if b.ndim != 2 or b.shape[1] != self.filt.order + 1:
            raise ValueError('order of the filter coefficients does not match the filter')
        ctx = self.ctx or _lib.default_context()
        y = np.empty_like(x)
        _lib.check(ctx, lib.kwy_mlsa_synthesis(ctx.handle, ptr(x), len(x), ptr(b), b.shape[0], self.filt.order,
                                               self.filt.alpha, self.filt.pd, self.hopsize, ptr(y)))
        return y
