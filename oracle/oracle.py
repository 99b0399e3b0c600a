"""ctypes front-end of the CPU oracle (``oracle/liboracle.so``).

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; nothing under
``kwiiyatta_amd/`` does.  The functions mirror the third-party signatures the
reference calls (pyworld / pysptk / fastdtw / nnmnkwii), so the parity tests
read like the reference's own tests:

  pyworld.dio/stonemask/cheaptrick/d4c/synthesize/get_cheaptrick_fft_size
      /root/reference/kwiiyatta/vocoder/world.py:35-96
  pysptk.sp2mc / mc2sp / util.mcepalpha
      /root/reference/kwiiyatta/vocoder/mcep.py:26,65,71
  fastdtw.fastdtw          /root/reference/kwiiyatta/vocoder/align.py:71
  nnmnkwii delta_features / MLPG.transform
      /root/reference/kwiiyatta/converter/delta.py:30,46, gmm.py:28-34

Parity status: these are restatements of the published algorithms, pinned by
the reference's statistical KAT envelopes only ("sample-level parity with
pyworld/pysptk/fastdtw/nnmnkwii unpinned"), see tests/test_oracle_kat.py.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'liboracle.so')

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_int64_p = ctypes.POINTER(ctypes.c_int64)

KO_MAX_WIN = 8

DELTA_WINDOWS = [
    (0, 0, np.array([1.0])),
    (1, 1, np.array([-0.5, 0.0, 0.5])),
    (1, 1, np.array([1.0, -2.0, 1.0])),
]


def build(force=False):
    """Compile liboracle.so with oracle/Makefile (gcc)."""
    if force or not os.path.exists(_SO):
        subprocess.check_call(['make', '-C', _HERE, '-s'] + (['-B'] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.ko_cheaptrick_f0_floor.restype = ctypes.c_double
        L.ko_dio_samples.restype = ctypes.c_int64
        L.ko_synth_timebase.restype = ctypes.c_int64
        L.ko_cheaptrick_f0_floor.argtypes = [ctypes.c_int, ctypes.c_int]
        L.ko_cheaptrick_fft_size.argtypes = [ctypes.c_int, ctypes.c_double]
        L.ko_dio_samples.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_double]
        L.ko_randn_fill.argtypes = [c_double_p, ctypes.c_int64]
        L.ko_dio.argtypes = [c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                             ctypes.c_double, ctypes.c_double, ctypes.c_double,
                             ctypes.c_int, ctypes.c_double, c_double_p, c_double_p]
        L.ko_stonemask.argtypes = [c_double_p, ctypes.c_int64, ctypes.c_int, c_double_p,
                                   c_double_p, ctypes.c_int64, c_double_p]
        L.ko_cheaptrick.argtypes = [c_double_p, ctypes.c_int64, ctypes.c_int, c_double_p,
                                    c_double_p, ctypes.c_int64, ctypes.c_double,
                                    ctypes.c_double, ctypes.c_int, c_double_p]
        L.ko_d4c.argtypes = [c_double_p, ctypes.c_int64, ctypes.c_int, c_double_p,
                             c_double_p, ctypes.c_int64, ctypes.c_double, ctypes.c_int,
                             c_double_p]
        L.ko_synth_timebase.argtypes = [c_double_p, ctypes.c_int64, ctypes.c_int,
                                        ctypes.c_double, ctypes.c_int64, ctypes.c_int,
                                        c_int32_p, c_double_p, c_double_p]
        L.ko_synthesize.argtypes = [c_double_p, ctypes.c_int64, c_double_p, c_double_p,
                                    ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                    ctypes.c_int64, c_double_p]
        L.ko_freqt.argtypes = [c_double_p, ctypes.c_int, c_double_p, ctypes.c_int,
                               ctypes.c_double]
        L.ko_sp2mc.argtypes = [c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                               ctypes.c_double, c_double_p]
        L.ko_mc2sp.argtypes = [c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                               ctypes.c_int, c_double_p]
        L.ko_fastdtw.argtypes = [c_double_p, ctypes.c_int64, c_double_p, ctypes.c_int64,
                                 ctypes.c_int, ctypes.c_int, c_double_p, c_int32_p,
                                 c_int64_p]
        L.ko_delta_features.argtypes = [c_double_p, ctypes.c_int64, ctypes.c_int,
                                        ctypes.c_int, c_int_p, c_int_p, c_double_p,
                                        c_double_p]
        L.ko_mlpg.argtypes = [c_double_p, c_double_p, ctypes.c_int64, ctypes.c_int,
                              ctypes.c_int, c_int_p, c_int_p, c_double_p, c_double_p]
        L.ko_gmm_mlpg.argtypes = [c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                  c_double_p, c_double_p, c_double_p, ctypes.c_int,
                                  ctypes.c_int, c_int_p, c_int_p, c_double_p, c_double_p,
                                  c_int32_p]
        L.ko_d4c_num_bands.argtypes = [ctypes.c_int]
        L.ko_code_aperiodicity.argtypes = [c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, c_double_p]
        L.ko_decode_aperiodicity.argtypes = [c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, c_double_p]
        L.ko_mc2b.argtypes = [c_double_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, c_double_p]
        L.ko_mlsa_synthesis.argtypes = [c_double_p, ctypes.c_int64, c_double_p, ctypes.c_int64, ctypes.c_int,
                                        ctypes.c_double, ctypes.c_int, ctypes.c_int, c_double_p]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _chk(a, name='ndarray'):
    """pyworld's contract: float64, C-contiguous
    (/root/reference/tests/kwiiyatta/vocoder/test_world.py:21-40)."""
    a = np.asarray(a)
    if not a.flags['C_CONTIGUOUS']:
        raise ValueError('ndarray is not C-contiguous')
    if a.dtype != np.float64:
        raise ValueError("Buffer dtype mismatch, expected 'double'")
    return a


default_frame_period = 5.0
default_f0_floor = 71.0
default_f0_ceil = 800.0


def get_cheaptrick_fft_size(fs, f0_floor=default_f0_floor):
    return lib().ko_cheaptrick_fft_size(int(fs), float(f0_floor))


def get_cheaptrick_f0_floor(fs, fft_size):
    return lib().ko_cheaptrick_f0_floor(int(fs), int(fft_size))


def randn(n):
    out = np.empty(n)
    lib().ko_randn_fill(_dp(out), n)
    return out


def dio(x, fs, f0_floor=default_f0_floor, f0_ceil=default_f0_ceil,
        channels_in_octave=2.0, frame_period=default_frame_period, speed=1,
        allowed_range=0.1):
    x = _chk(x)
    T = lib().ko_dio_samples(int(fs), len(x), float(frame_period))
    f0 = np.zeros(T)
    t = np.zeros(T)
    rc = lib().ko_dio(_dp(x), len(x), int(fs), f0_floor, f0_ceil, channels_in_octave,
                      float(frame_period), int(speed), allowed_range, _dp(t), _dp(f0))
    if rc != 0:
        raise ValueError('oracle dio: unsupported option (speed must be 1)')
    return f0, t


def stonemask(x, f0, temporal_positions, fs):
    x, f0, t = _chk(x), _chk(f0), _chk(temporal_positions)
    out = np.zeros(len(f0))
    lib().ko_stonemask(_dp(x), len(x), int(fs), _dp(t), _dp(f0), len(f0), _dp(out))
    return out


def cheaptrick(x, f0, temporal_positions, fs, q1=-0.15, f0_floor=default_f0_floor,
               fft_size=None):
    x, f0, t = _chk(x), _chk(f0), _chk(temporal_positions)
    if fft_size is None:
        fft_size = get_cheaptrick_fft_size(fs, f0_floor)
    out = np.zeros((len(f0), fft_size // 2 + 1))
    lib().ko_cheaptrick(_dp(x), len(x), int(fs), _dp(t), _dp(f0), len(f0), q1,
                        f0_floor, int(fft_size), _dp(out))
    return out


def d4c(x, f0, temporal_positions, fs, threshold=0.85, fft_size=None):
    x, f0, t = _chk(x), _chk(f0), _chk(temporal_positions)
    if fft_size is None:
        fft_size = get_cheaptrick_fft_size(fs, default_f0_floor)
    out = np.zeros((len(f0), fft_size // 2 + 1))
    lib().ko_d4c(_dp(x), len(x), int(fs), _dp(t), _dp(f0), len(f0), threshold,
                 int(fft_size), _dp(out))
    return out


def code_aperiodicity(aperiodicity, fs):
    """pyworld.code_aperiodicity: (T, K) -> (T, number of 3 kHz bands)"""
    ap = _chk(aperiodicity)
    nb = lib().ko_d4c_num_bands(int(fs))
    out = np.zeros((ap.shape[0], nb))
    lib().ko_code_aperiodicity(_dp(ap), ap.shape[0], int(fs), 2 * (ap.shape[1] - 1), _dp(out))
    return out


def decode_aperiodicity(coded_aperiodicity, fs, fft_size):
    """pyworld.decode_aperiodicity: (T, bands) -> (T, fft_size/2+1)"""
    c = _chk(coded_aperiodicity)
    out = np.zeros((c.shape[0], fft_size // 2 + 1))
    lib().ko_decode_aperiodicity(_dp(c), c.shape[0], int(fs), int(fft_size), c.shape[1], _dp(out))
    return out


def synth_timebase(f0, fs, frame_period, y_length, fft_size):
    f0 = _chk(f0)
    idx = np.zeros(y_length, dtype=np.int32)
    shift = np.zeros(y_length)
    vuv = np.zeros(y_length)
    n = lib().ko_synth_timebase(_dp(f0), len(f0), int(fs), float(frame_period),
                                int(y_length), int(fft_size),
                                idx.ctypes.data_as(c_int32_p), _dp(shift), _dp(vuv))
    return idx[:n].copy(), shift[:n].copy(), vuv


def synthesize(f0, spectrogram, aperiodicity, fs, frame_period=default_frame_period):
    f0, sp, ap = _chk(f0), _chk(spectrogram), _chk(aperiodicity)
    y_length = int(len(f0) * frame_period * fs / 1000)
    fft_size = (sp.shape[1] - 1) * 2
    y = np.zeros(y_length)
    lib().ko_synthesize(_dp(f0), len(f0), _dp(sp), _dp(ap), fft_size, float(frame_period),
                        int(fs), y_length, _dp(y))
    return y


def freqt(c, order, alpha):
    c = np.ascontiguousarray(c, dtype=np.float64)
    out = np.zeros(order + 1)
    lib().ko_freqt(_dp(c), len(c) - 1, _dp(out), order, alpha)
    return out


def _pow2(n):
    return n >= 2 and (n & (n - 1)) == 0


def sp2mc(powerspec, order, alpha):
    """pysptk.sp2mc: c = np.fft.irfft(log P); c[0] /= 2; freqt(c, order, alpha), row-wise.  Power-of-two
    transform lengths run in C (ko_sp2mc); any other even length (the reference's resampled spectra,
    kwiiyatta/vocoder/mcep.py:31-45) uses numpy's irfft -- upstream's own transform -- and the C freqt."""
    sp = np.ascontiguousarray(powerspec, dtype=np.float64)
    one = sp.ndim == 1
    sp2 = np.atleast_2d(sp)
    mc = np.zeros((sp2.shape[0], order + 1))
    if _pow2(2 * (sp2.shape[1] - 1)):
        lib().ko_sp2mc(_dp(sp2), sp2.shape[0], sp2.shape[1], order, alpha, _dp(mc))
    else:
        c = np.ascontiguousarray(np.fft.irfft(np.log(sp2), axis=1))
        c[:, 0] /= 2.0
        for t in range(len(c)):
            lib().ko_freqt(_dp(c[t]), c.shape[1] - 1, _dp(mc[t]), order, alpha)
    return mc[0] if one else mc


def mc2sp(mc, alpha, fftlen):
    """pysptk.mc2sp: c = freqt(mc, fftlen/2, -alpha); c[0] *= 2; mirror; exp(np.fft.rfft(c).real)."""
    mc = np.ascontiguousarray(mc, dtype=np.float64)
    one = mc.ndim == 1
    mc2 = np.atleast_2d(mc)
    half = int(fftlen) // 2
    sp = np.zeros((mc2.shape[0], half + 1))
    if _pow2(int(fftlen)):
        lib().ko_mc2sp(_dp(mc2), mc2.shape[0], mc2.shape[1] - 1, alpha, int(fftlen), _dp(sp))
    else:
        c = np.zeros(half + 1)
        symc = np.zeros((mc2.shape[0], int(fftlen)))
        for t in range(len(mc2)):
            lib().ko_freqt(_dp(mc2[t]), mc2.shape[1] - 1, _dp(c), half, -alpha)
            symc[t, 0] = 2.0 * c[0]
            symc[t, 1:half + 1] = c[1:]
            symc[t, -1:-half - 1:-1] = c[1:]
        sp = np.exp(np.fft.rfft(symc, axis=1).real)
    return sp[0] if one else sp


def mc2b(mc, alpha):
    """pysptk.mc2b (rows of a 2-D array independently)"""
    mc = np.ascontiguousarray(mc, dtype=np.float64)
    m2 = np.atleast_2d(mc)
    out = np.zeros_like(m2)
    lib().ko_mc2b(_dp(m2), m2.shape[0], m2.shape[1] - 1, float(alpha), _dp(out))
    return out.reshape(mc.shape)


def mlsa_synthesis(source, b, alpha, hopsize, pd=4):
    """pysptk.synthesis.Synthesizer(MLSADF(order, alpha, pd), hopsize).synthesis(source, b)"""
    x = np.ascontiguousarray(source, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    y = np.zeros_like(x)
    rc = lib().ko_mlsa_synthesis(_dp(x), len(x), _dp(b), b.shape[0], b.shape[1] - 1, float(alpha), int(pd),
                                 int(hopsize), _dp(y))
    if rc != 0:
        raise ValueError('mlsa_synthesis: bad argument')
    return y


def mcepalpha(fs, start=0.0, stop=1.0, step=0.001, num_points=1000):
    """pysptk.util.mcepalpha restated in numpy (host-side, trivial)."""
    alphas = np.arange(start, stop, step)
    mstep = (fs / 2.0) / num_points
    mel = 1000.0 / np.log(2) * np.log(1 + mstep * np.arange(0, num_points) / 1000.0)
    mel = mel / mel[-1]
    omega = np.pi / num_points * np.arange(0, num_points)
    best, best_a = None, None
    for a in alphas:
        num = (1 - a * a) * np.sin(omega)
        den = (1 + a * a) * np.cos(omega) - 2 * a
        with np.errstate(divide='ignore', invalid='ignore'):
            w = np.arctan(num / den)
        w[w < 0] += np.pi
        w = w / w[-1]
        d = np.sqrt(np.mean((mel - w) ** 2))
        if best is None or d < best:
            best, best_a = d, a
    return best_a


def fastdtw(x, y, radius=1, dist=2):
    assert dist == 2, 'oracle restates the dist=2 (Euclidean) call only'
    if int(radius) < 1:        # fastdtw 0.3.2 itself fails there (a KeyError: the last odd row gets no window)
        raise ValueError('fastdtw: radius must be >= 1')
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    if x.ndim == 1:
        x = x[:, None]
        y = y[:, None]
    d = ctypes.c_double()
    n = ctypes.c_int64()
    path = np.zeros((len(x) + len(y) + 1, 2), dtype=np.int32)
    lib().ko_fastdtw(_dp(x), len(x), _dp(y), len(y), x.shape[1], int(radius),
                     ctypes.byref(d), path.ctypes.data_as(c_int32_p), ctypes.byref(n))
    return d.value, [(int(a), int(b)) for a, b in path[:n.value]]


def _windows(windows):
    nwin = len(windows)
    wl = (ctypes.c_int * nwin)(*[w[0] for w in windows])
    wu = (ctypes.c_int * nwin)(*[w[1] for w in windows])
    coef = np.zeros((nwin, KO_MAX_WIN))
    for i, w in enumerate(windows):
        coef[i, :len(w[2])] = w[2]
    return nwin, wl, wu, coef


def delta_features(x, windows):
    x = np.ascontiguousarray(x, dtype=np.float64)
    nwin, wl, wu, coef = _windows(windows)
    out = np.zeros((x.shape[0], x.shape[1] * nwin))
    lib().ko_delta_features(_dp(x), x.shape[0], x.shape[1], nwin, wl, wu, _dp(coef), _dp(out))
    return out


def mlpg(mean_frames, variance_frames, windows):
    E = np.ascontiguousarray(mean_frames, dtype=np.float64)
    D = np.ascontiguousarray(variance_frames, dtype=np.float64)
    nwin, wl, wu, coef = _windows(windows)
    sd = E.shape[1] // nwin
    y = np.zeros((E.shape[0], sd))
    lib().ko_mlpg(_dp(E), _dp(D), E.shape[0], sd, nwin, wl, wu, _dp(coef), _dp(y))
    return y


def gmm_mlpg(x, weights, means, covs, windows=DELTA_WINDOWS, diff=False, return_mix=False):
    """MLPG(gmm, windows, diff).transform(delta_features(x, windows))[:, :d]."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    weights = np.ascontiguousarray(weights, dtype=np.float64)
    means = np.ascontiguousarray(means, dtype=np.float64)
    covs = np.ascontiguousarray(covs, dtype=np.float64)
    nwin, wl, wu, coef = _windows(windows)
    T, d = x.shape
    M = len(weights)
    assert means.shape == (M, 2 * nwin * d) and covs.shape == (M, 2 * nwin * d, 2 * nwin * d)
    y = np.zeros((T, d))
    mix = np.zeros(T, dtype=np.int32)
    rc = lib().ko_gmm_mlpg(_dp(x), T, d, M, _dp(weights), _dp(means), _dp(covs), int(diff),
                           nwin, wl, wu, _dp(coef), _dp(y), mix.ctypes.data_as(c_int32_p))
    if rc != 0:
        raise ValueError('oracle gmm_mlpg: covariance not positive definite')
    return (y, mix) if return_mix else y


def stretch_log(rows, new_bins, edge_periods=20):
    """Synthesizer._reshape_feature on log values (/root/reference/kwiiyatta/vocoder/abc/synthesizer.py:31-54) with the
    reference's own third-party call, scipy.signal.resample_poly (scipy is installed here): the checker of
    kwy_stretch_log."""
    import math
    import scipy.signal
    rows = np.asarray(rows, dtype=np.float64)
    bins = rows.shape[1]
    unit = math.gcd(bins, int(new_bins))
    lead_in, lead_out = bins // unit * edge_periods, int(new_bins) // unit * edge_periods
    logs = np.log(rows)
    wide = np.hstack((np.repeat(logs[:, :1], lead_in, axis=1), logs, np.repeat(logs[:, -1:], lead_in, axis=1)))
    wide = scipy.signal.resample_poly(wide, int(new_bins), bins, axis=1)
    return np.exp(wide[:, lead_out:wide.shape[1] - lead_out])


def gmm_convert_frames(x, weights, means, covs, diff=False):
    """nnmnkwii.baseline.gmm.MLPGBase.transform (0.0.17) -- what MLPG(gmm, windows=DELTA_WINDOWS[0:1], diff)
    .transform(X) runs, i.e. GMMFeatureConverter.convert(mlpg=False) (/root/reference/kwiiyatta/converter/gmm.py:28-34):
    per frame the posterior-weighted conditional mean, posteriors from scikit-learn's own predict_proba."""
    import sklearn.mixture
    from sklearn.mixture._gaussian_mixture import _compute_precision_cholesky
    x = np.asarray(x, dtype=np.float64)
    M, D = len(weights), x.shape[1]
    src_means, tgt_means = means[:, :D], means[:, D:]
    cxx, cxy, cyx = covs[:, :D, :D], covs[:, :D, D:], covs[:, D:, :D]
    if diff:
        tgt_means = tgt_means - src_means
        cxy = cxy - cxx
        cyx = np.transpose(cxy, (0, 2, 1))
    px = sklearn.mixture.GaussianMixture(n_components=M, covariance_type='full')
    px.weights_, px.means_, px.covariances_ = weights, src_means, cxx
    px.precisions_cholesky_ = _compute_precision_cholesky(cxx, 'full')
    post = px.predict_proba(x)
    out = np.zeros_like(x)
    for t in range(len(x)):
        E = np.empty((M, D))
        for m in range(M):
            E[m] = tgt_means[m] + cyx[m].dot(np.linalg.solve(cxx[m], x[t] - src_means[m]))
        out[t] = post[t].dot(E)
    return out
