"""pyworld-shaped front-end of the HIP WORLD kernels.

Same names, argument meaning, defaults and error behaviour as the pyworld 0.2.8
functions the reference calls (/root/reference/kwiiyatta/vocoder/world.py:35-96):
float64 C-contiguous inputs (``ValueError('ndarray is not C-contiguous')``
otherwise), freshly allocated numpy outputs.
"""
import numpy as np

from .. import _lib
from .._lib import lib, ptr

default_frame_period = 5.0
default_f0_floor = 71.0
default_f0_ceil = 800.0


def get_cheaptrick_fft_size(fs, f0_floor=default_f0_floor):
    return lib.kwy_cheaptrick_fft_size(int(fs), float(f0_floor))


def get_cheaptrick_f0_floor(fs, fft_size):
    return 3.0 * fs / (fft_size - 3.0)


def cheaptrick(x, f0, temporal_positions, fs, q1=-0.15, f0_floor=default_f0_floor,
               fft_size=None, ctx=None, out_div=1.0):
    x = _lib.as_f64(x)
    f0 = _lib.as_f64(f0)
    t = _lib.as_f64(temporal_positions)
    if len(t) != len(f0):
        raise ValueError('f0 and temporal_positions must have the same length')
    if fft_size is None:
        fft_size = get_cheaptrick_fft_size(fs, f0_floor)
    ctx = ctx or _lib.default_context()
    out = np.empty((len(f0), fft_size // 2 + 1))
    _lib.check(ctx, lib.kwy_cheaptrick(ctx.handle, ptr(x), len(x), int(fs), ptr(t), ptr(f0),
                                       len(f0), float(q1), float(f0_floor), int(fft_size),
                                       float(out_div), ptr(out)))
    return out


def d4c(x, f0, temporal_positions, fs, threshold=0.85, fft_size=None, ctx=None):
    x = _lib.as_f64(x)
    f0 = _lib.as_f64(f0)
    t = _lib.as_f64(temporal_positions)
    if len(t) != len(f0):
        raise ValueError('f0 and temporal_positions must have the same length')
    if fft_size is None:
        fft_size = get_cheaptrick_fft_size(fs, default_f0_floor)
    ctx = ctx or _lib.default_context()
    out = np.empty((len(f0), fft_size // 2 + 1))
    _lib.check(ctx, lib.kwy_d4c(ctx.handle, ptr(x), len(x), int(fs), ptr(t), ptr(f0), len(f0),
                                float(threshold), int(fft_size), ptr(out)))
    return out


def synthesize(f0, spectrogram, aperiodicity, fs, frame_period=default_frame_period, ctx=None,
               sp_mul=1.0):
    f0 = _lib.as_f64(f0)
    sp = _lib.as_f64(spectrogram)
    ap = _lib.as_f64(aperiodicity)
    if sp.ndim != 2 or sp.shape != ap.shape or sp.shape[0] != len(f0):
        raise ValueError('f0, spectrogram and aperiodicity shapes do not match')
    fft_size = (sp.shape[1] - 1) * 2
    y_length = int(len(f0) * frame_period * fs / 1000)
    ctx = ctx or _lib.default_context()
    y = np.empty(y_length)
    _lib.check(ctx, lib.kwy_synthesize(ctx.handle, ptr(f0), len(f0), ptr(sp), ptr(ap), fft_size,
                                       float(frame_period), int(fs), float(sp_mul), y_length,
                                       ptr(y)))
    return y


def get_num_aperiodicities(fs):
    return lib.kwy_aperiodicity_bands(int(fs))


def code_aperiodicity(aperiodicity, fs, ctx=None):
    """(T, K) aperiodicity -> (T, bands) dB values at 3 kHz, 6 kHz, ...  (pyworld.code_aperiodicity)"""
    ap = _lib.as_f64(aperiodicity)
    if ap.ndim != 2:
        raise ValueError('aperiodicity must be 2-dimensional')
    ctx = ctx or _lib.default_context()
    nb = get_num_aperiodicities(fs)
    coded = np.empty((ap.shape[0], nb))
    _lib.check(ctx, lib.kwy_code_aperiodicity(ctx.handle, ptr(ap), ap.shape[0], int(fs),
                                              (ap.shape[1] - 1) * 2, ptr(coded)))
    return coded


def decode_aperiodicity(coded_aperiodicity, fs, fft_size, ctx=None):
    """(T, bands) -> (T, fft_size/2+1)  (pyworld.decode_aperiodicity)"""
    coded = _lib.as_f64(coded_aperiodicity)
    if coded.ndim != 2:
        raise ValueError('coded_aperiodicity must be 2-dimensional')
    ctx = ctx or _lib.default_context()
    ap = np.empty((coded.shape[0], fft_size // 2 + 1))
    _lib.check(ctx, lib.kwy_decode_aperiodicity(ctx.handle, ptr(coded), coded.shape[0], int(fs), int(fft_size),
                                                coded.shape[1], ptr(ap)))
    return ap


def dio(x, fs, f0_floor=default_f0_floor, f0_ceil=default_f0_ceil, channels_in_octave=2.0,
        frame_period=default_frame_period, speed=1, allowed_range=0.1, ctx=None):
    x = _lib.as_f64(x)
    ctx = ctx or _lib.default_context()
    T = lib.kwy_dio_frames(int(fs), len(x), float(frame_period))
    f0 = np.empty(T)
    t = np.empty(T)
    _lib.check(ctx, lib.kwy_dio(ctx.handle, ptr(x), len(x), int(fs), float(f0_floor), float(f0_ceil),
                                float(channels_in_octave), float(frame_period), int(speed),
                                float(allowed_range), ptr(t), ptr(f0)))
    return f0, t


def stonemask(x, f0, temporal_positions, fs, ctx=None):
    x = _lib.as_f64(x)
    f0 = _lib.as_f64(f0)
    t = _lib.as_f64(temporal_positions)
    if len(t) != len(f0):
        raise ValueError('f0 and temporal_positions must have the same length')
    ctx = ctx or _lib.default_context()
    out = np.empty(len(f0))
    _lib.check(ctx, lib.kwy_stonemask(ctx.handle, ptr(x), len(x), int(fs), ptr(t), ptr(f0), len(f0),
                                      ptr(out)))
    return out


# ---- the f0 track of a batch of utterances without the host (device tensors; round 5) -------------------------------
def dio_frames(fs, x_length, frame_period=default_frame_period):
    return int(lib.kwy_dio_frames(int(fs), int(x_length), float(frame_period)))


def dio_batch_dev(ctx, waves, fs, t_out, f0_out, status=None, f0_floor=default_f0_floor, f0_ceil=default_f0_ceil,
                  channels_in_octave=2.0, frame_period=default_frame_period, speed=1, allowed_range=0.1):
    """pyworld.dio for every waveform of `waves` (float64 device tensors), enqueued on the context's stream and not
    synchronised.  t_out / f0_out: per utterance a device tensor of dio_frames(...) values; status: an int32 device
    tensor with one word per utterance (non-zero afterwards: zero-crossing buffer overflow), or None."""
    rows = [(x, x.numel(), t, f, (status[i:i + 1] if status is not None else 0))
            for i, (x, t, f) in enumerate(zip(waves, t_out, f0_out))]
    jobs = _lib.job_array(_lib.F0Job, rows)
    _lib.check(ctx, lib.kwy_dio_batch_dev(ctx.handle, jobs, len(rows), int(fs), float(f0_floor), float(f0_ceil),
                                          float(channels_in_octave), float(frame_period), int(speed),
                                          float(allowed_range)))


def stonemask_batch_dev(ctx, waves, t, f0, fs, out):
    """pyworld.stonemask for every utterance (device tensors), one grid over all frames; not synchronised"""
    arr = _lib.utterance_array(list(zip(waves, t, f0, out)))
    _lib.check(ctx, lib.kwy_stonemask_batch_dev(ctx.handle, arr, len(arr), int(fs)))
