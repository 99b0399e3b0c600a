"""The spectral-axis polyphase stretch on the GPU (kwy_stretch.hip): the `scipy.signal.resample_poly` call of the
reference's Synthesizer._reshape_feature (/root/reference/kwiiyatta/vocoder/abc/synthesizer.py:31-44) together with
the edge replication, log and exp around it."""
import numpy as np

from .. import _lib
from .._lib import lib, ptr


def stretch_log(rows, new_bins, ctx=None):
    """(T, K) positive values -> (T, new_bins): exp of the rationally resampled logarithms, edges replicated for
    20 filter periods and cut off again (reshape_spectrum_envelope / reshape_aperiodicity)."""
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    if rows.ndim != 2:
        raise ValueError('stretch_log expects a (frames, bins) matrix')
    ctx = ctx or _lib.default_context()
    out = np.empty((rows.shape[0], int(new_bins)))
    if rows.shape[0] == 0:
        return out
    _lib.check(ctx, lib.kwy_stretch_log(ctx.handle, ptr(rows), rows.shape[0], rows.shape[1], int(new_bins), ptr(out)))
    return out
