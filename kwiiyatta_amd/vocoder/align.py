"""DTW alignment of two features (mirrors
/root/reference/kwiiyatta/vocoder/align.py:10-146).  The DTW itself runs on the
GPU through kwiiyatta_amd.backend.dtw (fastdtw-shaped); the feature
construction and the path projection are host-side index bookkeeping."""
import numpy as np

import kwiiyatta_amd as kwiiyatta
from ..backend import dtw as fastdtw


def binalize(x, threshold, ceil, floor=0, out=None):
    if out is None:
        out = np.full_like(x, floor)
    else:
        out[:] = floor
    out[x >= threshold] = ceil
    return out


_POWER_PIVOTS = {
    'max': lambda p, thr: p.max() - thr,
    'median': lambda p, thr: np.median(p) - thr,
    'min': lambda p, thr: p.min() + thr,
    'fix': lambda p, thr: thr,
}


def make_feature(f, fs, vuv='voiced', vuv_weight=9.0, power='binalize', power_weight=9.4,
                 power_pivot='max', power_threshold=1.636):
    """DTW feature rows: [power term, voicing term, mc1 .. mcN]."""
    mc = f.resample_mel_cepstrum(fs).data
    power_track = mc[:, 0]
    out = np.hstack((np.zeros((len(mc), 2)), mc[:, 1:]))

    if power == 'binalize':
        if power_pivot not in _POWER_PIVOTS:
            raise ValueError(f'Unknown power_pivot parameter: {power_pivot!r}')
        binalize(power_track, _POWER_PIVOTS[power_pivot](power_track, power_threshold),
                 power_weight, out=out[:, 0])
    elif power == 'raw':
        out[:, 0] = power_track
    elif power is not None:
        raise ValueError(f'Unknown power parameter: {power!r}')

    if vuv == 'voiced':
        out[:, 1][f.is_voiced] = vuv_weight
    elif vuv == 'f0':
        out[:, 1][f.f0 > 0] = vuv_weight
    elif vuv is not None:
        raise ValueError(f'Unknown vuv parameter: {vuv!r}')
    return out


def dtw_feature(x, y, vuv='voiced', power='binalize', strict=True, radius=32, **kwargs):
    fs = min(x.fs, y.fs)
    kwargs.update(vuv=vuv, power=power)
    x_feature = make_feature(x, fs, **kwargs)
    y_feature = make_feature(y, fs, **kwargs)

    dist, path = fastdtw.fastdtw(x_feature, y_feature, dist=2, radius=radius)

    def consistent(i, j):
        # (sic) the voicing test compares x's voicing column with y's POWER column,
        # exactly as the reference does (align.py:78)
        if power == 'binalize' and ((x_feature[i, 0] > 0) ^ (y_feature[j, 0] > 0)):
            return False
        if vuv is not None and ((x_feature[i, 1] > 0) ^ (y_feature[j, 0] > 0)):
            return False
        return True

    if strict:
        kept = [path[0]] + [(i, j) for i, j in path[1:-1] if consistent(i, j)] + [path[-1]]
        path = np.array(kept, dtype=int).reshape((-1, 2))
    else:
        path = np.array(path)
    return dist, path


def project_path_iter(path, trim=True, trim_len=1):
    """Walk a DTW path and yield, for every y index (pads trimmed), one x index."""
    prev_x = prev_y = -1
    len_y = path[-1][1] + 1
    if trim:
        prev_y += trim_len
        len_y -= trim_len
    for x, y in path:
        if y <= prev_y:
            continue
        if y - prev_y > 1:                      # y jumped: spread x over the gap
            y = min(y, len_y - 1)
            diff_x, diff_y = x - prev_x, y - prev_y
            for i in range(diff_y):
                yield prev_x + diff_x * i // (diff_y - 1)
        elif y >= len_y:
            break
        else:
            yield x
        prev_x, prev_y = x, y


def align(feature, target, vuv='f0', strict=False, pad_silence=True, pad_len=100, **kwargs):
    """`feature` re-timed onto `target`'s frame axis."""
    if pad_silence:
        feature = kwiiyatta.pad_silence(feature, frame_len=pad_len)
        target = kwiiyatta.pad_silence(target, frame_len=pad_len)
    _, path = dtw_feature(feature, target, vuv=vuv, strict=strict, **kwargs)
    return feature[list(project_path_iter(path, trim=pad_silence, trim_len=pad_len))]


def align_even(a, b, pad_silence=True, pad_len=100, **kwargs):
    """Both features re-timed onto the common DTW path."""
    if pad_silence:
        a = kwiiyatta.pad_silence(a, pad_len)
        b = kwiiyatta.pad_silence(b, pad_len)
    _, path = dtw_feature(a, b, **kwargs)
    path = np.array(path).T
    if pad_silence:
        begin = np.argmax(np.logical_and(path[0] >= pad_len, path[1] >= pad_len))
        end = np.argmax(np.logical_and(path[0] >= a.frame_len - pad_len,
                                       path[1] >= b.frame_len - pad_len))
        path = path[:, begin:end]
    return a[path[0]], b[path[1]]
