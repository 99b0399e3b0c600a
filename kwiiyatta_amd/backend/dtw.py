"""fastdtw-shaped front-end of the HIP FastDTW kernels
(reference call site /root/reference/kwiiyatta/vocoder/align.py:71)."""
import ctypes

import numpy as np

from .. import _lib
from .._lib import lib, ptr


def fastdtw(x, y, radius=1, dist=2, ctx=None):
    """Approximate DTW of two feature sequences under the Euclidean frame
    distance.  Returns (distance, path) with path a list of (i, j) tuples,
    like fastdtw.fastdtw."""
    if dist != 2:
        raise NotImplementedError('only dist=2 (Euclidean norm), the value the reference uses, '
                                  'is implemented on the GPU')
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    if x.ndim == 1:
        x = x[:, None]
    if y.ndim == 1:
        y = y[:, None]
    if x.shape[1] != y.shape[1]:
        raise ValueError('second dimension of x and y must be the same')
    ctx = ctx or _lib.default_context()
    d = ctypes.c_double()
    n = ctypes.c_int64()
    path = np.empty((len(x) + len(y) + 2, 2), dtype=np.int32)
    _lib.check(ctx, lib.kwy_fastdtw(ctx.handle, ptr(x), len(x), ptr(y), len(y), x.shape[1],
                                    int(radius), ctypes.byref(d), ptr(path), ctypes.byref(n)))
    return d.value, [(int(a), int(b)) for a, b in path[:n.value]]
