"""Operations along the spectral axis of (frames, bins) feature matrices -- the host-side arithmetic behind
`reshape` (another number of bins at the same sampling rate) and `resample` (another sampling rate at the same
bin spacing).  Plain numpy / scipy functions, no state; the vocoder classes bind them to their features.

Behaviour pinned by the reference (kwiiyatta/vocoder/abc/synthesizer.py:31-113) and by its tests
(tests/kwiiyatta/test_vocoder.py:291-467): stretching is a polyphase resampling of the rows after replicating
the edge bins (20 output periods on either side, cut off again afterwards), and it is applied to LOG values;
lowering the sampling rate keeps the leading bins, raising it appends bins supplied by the vocoder.
"""
import math

import numpy as np
import scipy.signal

EDGE_PERIODS = 20


def bins_at_rate(n_bins, fs, new_fs):
    """number of bins that cover the same frequency spacing at another sampling rate (integer arithmetic)"""
    return n_bins * new_fs // fs


def stretch(rows, new_bins):
    """(T, K) -> (T, new_bins): rational resampling of every row with replicated edges"""
    bins = rows.shape[1]
    unit = math.gcd(bins, new_bins)
    lead_in, lead_out = bins // unit * EDGE_PERIODS, new_bins // unit * EDGE_PERIODS
    left = np.repeat(rows[:, :1], lead_in, axis=1)
    right = np.repeat(rows[:, -1:], lead_in, axis=1)
    wide = scipy.signal.resample_poly(np.hstack((left, rows, right)), new_bins, bins, axis=1)
    return wide[:, lead_out:wide.shape[1] - lead_out]


def stretch_log(rows, new_bins):
    """`stretch` in the log domain (power spectra, aperiodicity ratios)"""
    return np.exp(stretch(np.log(rows), new_bins))


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
