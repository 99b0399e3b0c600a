"""pysptk-shaped front-end of the HIP mel-cepstrum kernels
(reference call sites /root/reference/kwiiyatta/vocoder/mcep.py:26,65,71)."""
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
