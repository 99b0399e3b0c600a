"""Operations along the spectral axis of (frames, bins) feature matrices -- the host-side arithmetic behind
`reshape` (another number of bins at the same sampling rate) and `resample` (another sampling rate at the same
bin spacing).  Pure functions, no state; the vocoder classes bind them to their features.  The stretch itself runs
on the GPU (kwiiyatta_amd.backend.resample).

Behaviour pinned by the reference (kwiiyatta/vocoder/abc/synthesizer.py:31-113) and by its tests
(tests/kwiiyatta/test_vocoder.py:291-467): stretching is a polyphase resampling of the rows after replicating
the edge bins (20 output periods on either side, cut off again afterwards), and it is applied to LOG values;
lowering the sampling rate keeps the leading bins, raising it appends bins supplied by the vocoder.
"""
from ..backend import resample as _gpu

EDGE_PERIODS = 20


def bins_at_rate(n_bins, fs, new_fs):
    """number of bins that cover the same frequency spacing at another sampling rate (integer arithmetic)"""
    return n_bins * new_fs // fs


def stretch_log(rows, new_bins):
    """(T, K) -> (T, new_bins) in the log domain (power spectra, aperiodicity ratios): rational resampling of the
    logarithm of every row with replicated edges (EDGE_PERIODS filter periods on either side, cut off again
    afterwards) -- kwy_stretch_log on the GPU"""
    return _gpu.stretch_log(rows, new_bins)


def keep_low_band(rows, new_bins):
    return rows[:, :new_bins]


def change_rate(rows, fs, new_fs, widen, narrow=keep_low_band):
    """dispatch on the direction of the change; `widen(rows, fs, new_fs, new_bins)` / `narrow(rows, new_bins)`"""
    if new_fs == fs:
        return rows
    new_bins = bins_at_rate(rows.shape[1], fs, new_fs)
    if new_fs > fs:
        return widen(rows, fs, new_fs, new_bins)
    return narrow(rows, new_bins)
