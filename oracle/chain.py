"""The whole config-3 path (analyse -> pad -> sp2mc -> FastDTW -> project -> GMM + MLPG -> mc2sp -> synthesis) on the
CPU oracle, every stage fed by the oracle's OWN previous output -- the all-CPU chain the HIP pipeline's final waveform
is compared with (north star: "output within 1e-4 RMS of the CPU reference on identical inputs").

TEST INFRASTRUCTURE ONLY, like the rest of ``oracle/``: used by ``tests/`` and by the ``cpu_baseline`` leg of
``bench.py`` (which times it and reports the RMS against the pipeline's waveform), never by the product.

Flow restated: /root/reference/kwiiyatta/resynthesize_voice.py:46-79 (analyse, carrier alignment),
convert_voice.py:35-46 (convert + synthesise), vocoder/align.py:20-58,99-120 (DTW features, one source frame per
target frame), vocoder/world.py:148-161 (pad_silence), converter/mcep.py:47-61 (c0 kept, c1.. converted).
"""
import time

import numpy as np

from . import oracle as ko

EPS = 2.220446049250313e-16
PAD_LEN = 100
FRAME_PERIOD = 5.0


def draw_silence(rng, fs, K, frame_len=PAD_LEN):
    """one block of WorldSynthesizer._silence_spectrum_envelope: |N(0, EPS / fs)|, (frame_len, K)"""
    return np.abs(rng.normal(0, EPS / fs, (frame_len, K)))


def project_path(path, trim_len):
    """One source frame per target frame (kwiiyatta/vocoder/align.py:99-120 as `align` uses it): the first x of every
    new y, gaps in y filled by spreading the x range with integer arithmetic, the silence pads cut off the y axis."""
    out = []
    seen_x, seen_y = -1, trim_len - 1      # y values up to seen_y are done (the leading pad counts as done)
    y_end = path[-1][1] + 1 - trim_len     # first y value that is not produced
    for x, y in path:
        gap = y - seen_y
        if gap < 1:
            continue
        if gap == 1:
            if y >= y_end:
                break
            out.append(x)
        else:                               # y values were skipped: spread x over them
            y = min(y, y_end - 1)
            gap, rise = y - seen_y, x - seen_x
            out.extend(seen_x + rise * i // (gap - 1) for i in range(gap))
        seen_x, seen_y = x, y
    return out


def pair_chain(src, tgt, gmm_params, fs, silence, order=24, radius=32, projector=None):
    """src / tgt: (x, f0, timeaxis); gmm_params: (weights, means, covariances) of the joint GMM over 2 * 3 * order
    dims; silence: the four (PAD_LEN, K) pad spectra (source head, source tail, target head, target tail) -- the
    same blocks the pipeline was given.  projector: path -> source index per target frame (default: the package's
    own host logic is NOT imported here; pass kwiiyatta_amd.vocoder.align.project_path_iter to use it).
    Returns a dict with every intermediate the tests compare and 'seconds' (wall time of the chain)."""
    weights, means, covs = gmm_params
    alpha = ko.mcepalpha(fs)
    fft = ko.get_cheaptrick_fft_size(fs)
    K = fft // 2 + 1
    P = PAD_LEN
    t0 = time.perf_counter()
    sides = []
    for (x, f0, t), (head, tail) in zip((src, tgt), (silence[:2], silence[2:])):
        sp = ko.cheaptrick(x, f0, t, fs) / fs
        ap = ko.d4c(x, f0, t, fs)
        sp_pad = np.ascontiguousarray(np.concatenate((head, sp, tail)))
        ap_pad = np.concatenate((np.full((P, K), 1 - 1e-12), ap, np.full((P, K), 1 - 1e-12)))
        f0_pad = np.r_[np.zeros(P), f0, np.zeros(P)]
        mc = ko.sp2mc(sp_pad, order, alpha)
        feat = np.hstack((np.zeros((len(mc), 2)), mc[:, 1:]))
        feat[:, 0][mc[:, 0] >= mc[:, 0].max() - 1.636] = 9.4
        feat[:, 1][f0_pad > 0] = 9.0
        sides.append(dict(sp_pad=sp_pad, ap_pad=ap_pad, mc=mc, feat=feat))
    dist, path = ko.fastdtw(sides[0]['feat'], sides[1]['feat'], radius=radius, dist=2)
    if projector is None:
        idx = np.array(project_path(path, P), dtype=np.int64)
    else:
        idx = np.fromiter(projector(np.array(path), trim=True, trim_len=P), dtype=np.int64)
    mc_al = sides[0]['mc'][idx]
    ap_al = np.ascontiguousarray(sides[0]['ap_pad'][idx])
    y = ko.gmm_mlpg(np.ascontiguousarray(mc_al[:, 1:]), weights, means, covs)
    mc_conv = np.hstack((mc_al[:, :1], y))
    sp_conv = ko.mc2sp(mc_conv, alpha, fft)
    wave = ko.synthesize(tgt[1], np.ascontiguousarray(sp_conv * fs), ap_al, fs, FRAME_PERIOD)
    return dict(seconds=time.perf_counter() - t0, frames=len(src[1]), dist=dist, path=path, idx=idx, mc_al=mc_al,
                ap_al=ap_al, mc_conv=mc_conv, sp_conv=sp_conv, wave=wave, sides=sides)
