"""nnmnkwii-shaped front-end of the HIP converter-apply kernels
(reference call sites /root/reference/kwiiyatta/converter/delta.py:30,46 and
gmm.py:28-34): ``MLPG(gmm, windows, diff).transform(X)`` with the reference's
fixed DELTA_WINDOWS."""
import numpy as np

from .. import _lib
from .._lib import lib, ptr

DELTA_WINDOWS = [
    (0, 0, np.array([1.0])),
    (1, 1, np.array([-0.5, 0.0, 0.5])),
    (1, 1, np.array([1.0, -2.0, 1.0])),
]


def _is_delta_windows(windows):
    return (len(windows) == 3 and
            all(w[0] == r[0] and w[1] == r[1] and np.array_equal(np.asarray(w[2]), r[2])
                for w, r in zip(windows, DELTA_WINDOWS)))


def _is_static_window(windows):
    w = DELTA_WINDOWS[0]
    return len(windows) == 1 and windows[0][0] == w[0] and windows[0][1] == w[1] and \
        np.array_equal(np.asarray(windows[0][2]), w[2])


def delta_features(x, windows):
    """static -> static|delta|delta2 (np.correlate(..., 'same') per column).
    Host-side helper for dataset preparation (the fit path); the conversion
    kernel computes the same features on the device."""
    x = np.asarray(x)
    T, D = x.shape
    y = np.zeros((T, D * len(windows)), dtype=x.dtype)
    for i, (_, _, w) in enumerate(windows):
        for c in range(D):
            y[:, D * i + c] = np.correlate(x[:, c], w, mode='same')
    return y


class MLPG:
    """Maximum-likelihood parameter generation from a joint GMM
    (sklearn.mixture.GaussianMixture-like object with weights_, means_,
    covariances_ and covariance_type == 'full')."""

    def __init__(self, gmm, windows=None, swap=False, diff=False, ctx=None):
        if windows is None:
            windows = DELTA_WINDOWS
        self.framewise = _is_static_window(windows)     # DELTA_WINDOWS[0:1]: conversion without trajectory smoothing
        if not self.framewise and not _is_delta_windows(windows):
            raise NotImplementedError('only the reference DELTA_WINDOWS (or their static part alone) are implemented '
                                      'on the GPU')
        if swap:
            raise NotImplementedError('swap=True is not used by the reference and not implemented')
        assert gmm.covariance_type == 'full'
        self.weights = np.ascontiguousarray(gmm.weights_, dtype=np.float64)
        self.means = np.ascontiguousarray(gmm.means_, dtype=np.float64)
        self.covs = np.ascontiguousarray(gmm.covariances_, dtype=np.float64)
        self.diff = bool(diff)
        self.windows = windows
        self.num_mixtures = len(self.weights)
        self.static_dim = self.means.shape[1] // 2 // len(windows)
        self.ctx = ctx

    def transform(self, src):
        """src: (T, 3*d) static+delta features (as produced by delta_features) or
        (T, d) static features.  Returns (T, 3*d) with the generated static
        trajectory in the first d columns (nnmnkwii returns (T, d); the
        reference truncates to d either way: converter/delta.py:48-49)."""
        src = np.ascontiguousarray(src, dtype=np.float64)
        d = self.static_dim
        if self.framewise:
            # nnmnkwii: feature_dim == static_dim -> MLPGBase.transform, the posterior-weighted conditional mean of
            # every frame on its own (all of the mixture's dimensions, dynamic features included, are "static" here)
            if src.shape[1] != d:
                raise ValueError(f'feature dimension {src.shape[1]} does not match the GMM ({d})')
            ctx = self.ctx or _lib.default_context()
            y = np.empty_like(src)
            if len(src):
                _lib.check(ctx, lib.kwy_gmm_convert_frames(ctx.handle, ptr(src), src.shape[0], d, self.num_mixtures,
                                                           ptr(self.weights), ptr(self.means), ptr(self.covs),
                                                           int(self.diff), ptr(y)))
            return y
        if src.shape[1] not in (d, 3 * d):
            raise ValueError(f'feature dimension {src.shape[1]} does not match the GMM ({d})')
        x = np.ascontiguousarray(src[:, :d])
        ctx = self.ctx or _lib.default_context()
        y = np.empty((x.shape[0], d))
        _lib.check(ctx, lib.kwy_gmm_mlpg(ctx.handle, ptr(x), x.shape[0], d, self.num_mixtures,
                                         ptr(self.weights), ptr(self.means), ptr(self.covs),
                                         int(self.diff), ptr(y)))
        return y
