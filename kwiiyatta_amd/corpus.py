"""Corpus-level path (BASELINE config 5): a parallel corpus -> joint training matrix in HBM -> converter
fit -> batch conversion, without the features ever visiting the host.

What the reference does for `kwiiyatta --source A --target B files...`
(/root/reference/kwiiyatta/config.py:83-104, convert_voice.py:6-46):

  per pair    analyse both sides; TrimmedDataset; align_even (pad 100 silent frames, DTW features with
              vuv='voiced' and a binarised power term, FastDTW radius 32, strict path filter, cut to the
              un-padded stretch); mel-cepstra of the aligned frames without c0; delta features;
              np.hstack + remove_zeros_frames; rows appended to one array (make_dataset_to_array)
  fit         GaussianMixture(n_components, covariance_type='full', max_iter=100, random_state=seed)
  per file    analyse; convert the mel-cepstrum (delta + GMM posterior + MLPG, c0 kept); synthesise

Here every pair runs on its own HIP stream (`TrainPair`), ranks take contiguous blocks of pairs (the global
row order is the pair order whatever the number of ranks), the rows land in one device tensor that
`GaussianMixtureHIP.fit` uses as its shard, and the fitted model converts utterances stream-parallel
(`ConvertPipeline`).  The only collectives are those of the fit.

The silence padding of `align_even` draws from numpy's GLOBAL legacy generator in the reference
(kwiiyatta/vocoder/world.py:158-161, quirk kept): the draws are made on the host, in the reference's order
(source head, source tail, target head, target tail, pair after pair), and uploaded once per pair -- so this path
and the Python API path produce the same matrix under `np.random.seed`.
"""
import numpy as np
import torch

from . import _lib
from ._lib import lib, c_vp
from .pipeline import (PAD_LEN, POWER_THRESHOLD, POWER_WEIGHT, SAFE_GUARD_MINIMUM, VUV_WEIGHT, DeviceGMM, _Graphed,
                       draw_silence)

TRIM_EPS = 1e-7       # nnmnkwii trim_zeros_frames / remove_zeros_frames


def _p(t):
    return c_vp(t.data_ptr())


class _TrainSide:
    def __init__(self, x, f0, t, fs, K, order, dev):
        f64 = dict(dtype=torch.float64, device=dev)
        self.N, self.T = len(x), len(f0)
        up = lambda a: a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
        self.x, self.f0, self.t = up(x), up(f0), up(t)
        Tp = self.T + 2 * PAD_LEN
        self.sp_pad = torch.empty((Tp, K), **f64)
        self.ap_pad = torch.full((Tp, K), 1 - SAFE_GUARD_MINIMUM, **f64)
        self.f0_pad = torch.zeros(Tp, **f64)
        self.n_keep_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.n = None          # frames kept by TrimmedDataset (host int, after analyse + sync)

    @property
    def sp(self):
        return self.sp_pad[PAD_LEN:PAD_LEN + self.T]

    @property
    def ap(self):
        return self.ap_pad[PAD_LEN:PAD_LEN + self.T]


class TrainPair:
    """One parallel pair -> its rows of the training matrix, on one stream.

        p.analyse()   enqueue CheapTrick + D4C of both sides and the trim lengths
        p.align()     (reads the two trim lengths: one 16-byte D2H) enqueue everything else
        p.rows()      (synchronises) the pair's joint rows, an (n, 2*3*order) device tensor
    """

    def __init__(self, device_index, fs, source, target, order=24, radius=32, frame_period=5.0, stream=None,
                 silence=None, ctx=None, silence_ready=None):
        """silence: the four (100, K) pad spectra (source head, source tail, target head, target tail) as numpy arrays
        or device tensors (then `silence_ready`: an event after which they are valid); default: drawn here from
        numpy's global generator in that order"""
        self.dev = torch.device('cuda', device_index)
        self.fs, self.order, self.radius, self.frame_period = int(fs), int(order), int(radius), float(frame_period)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.dev)
        self.ctx = ctx if ctx is not None else _lib.Context(device_index, stream=self.stream.cuda_stream)
        self.fft = lib.kwy_cheaptrick_fft_size(self.fs, 71.0)
        self.K = self.fft // 2 + 1
        from .backend import sptk
        self.alpha = sptk.mcepalpha(self.fs)
        if silence is None:        # the reference's order of draws: source head, source tail, target head, tail
            silence = [draw_silence(self.fs, self.K) for _ in range(4)]
        with torch.cuda.stream(self.stream):
            self.src = _TrainSide(*source, self.fs, self.K, order, self.dev)
            self.tgt = _TrainSide(*target, self.fs, self.K, order, self.dev)
            if silence_ready is not None:
                self.stream.wait_event(silence_ready)
            self.silence = [s if torch.is_tensor(s) else torch.from_numpy(np.ascontiguousarray(s)).to(self.dev)
                            for s in silence]
            # tensors made on other streams (the upload helper's, the generator's) are used on this one: the allocator
            # must not hand their memory out again while this stream's work on them is still queued
            for t_ in (self.src.x, self.src.f0, self.src.t, self.tgt.x, self.tgt.f0, self.tgt.t, *self.silence):
                t_.record_stream(self.stream)
        self.frames = self.src.T
        self.n_rows = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.joint = None

    def _chk(self, rc):
        _lib.check(self.ctx, rc)

    def analyse(self):
        h, fs, fft, K = self.ctx.handle, self.fs, self.fft, self.K
        with torch.cuda.stream(self.stream):
            for s in (self.src, self.tgt):
                self._chk(lib.kwy_cheaptrick_dev(h, _p(s.x), s.N, fs, _p(s.t), _p(s.f0), s.T, -0.15, 71.0, fft,
                                                 float(fs), _p(s.sp)))
                self._chk(lib.kwy_d4c_dev(h, _p(s.x), s.N, fs, _p(s.t), _p(s.f0), s.T, 0.85, fft, _p(s.ap)))
                self._chk(lib.kwy_trim_length_dev(h, _p(s.sp), s.T, K, TRIM_EPS, _p(s.n_keep_dev)))

    def align(self):
        h, fs, K, order, P = self.ctx.handle, self.fs, self.K, self.order, PAD_LEN
        f64 = dict(dtype=torch.float64, device=self.dev)
        i32 = dict(dtype=torch.int32, device=self.dev)
        with torch.cuda.stream(self.stream):
            keep = torch.cat((self.src.n_keep_dev, self.tgt.n_keep_dev)).cpu().tolist()   # waits for the analysis
            for s, n, sil in ((self.src, keep[0], self.silence[:2]), (self.tgt, keep[1], self.silence[2:])):
                s.n = int(n)
                s.Tp = s.n + 2 * P
                # pad_silence on the first n frames (TrimmedDataset keeps feature[:n])
                s.sp_pad[:P].copy_(sil[0])
                s.sp_pad[P + s.n:s.Tp].copy_(sil[1])
                s.ap_pad[P + s.n:s.Tp].fill_(1 - SAFE_GUARD_MINIMUM)
                s.f0_pad[P:P + s.n].copy_(s.f0[:s.n])
                s.voiced = torch.empty(s.Tp, **f64)
                s.mc_pad = torch.empty((s.Tp, order + 1), **f64)
                s.feat = torch.empty((s.Tp, order + 2), **f64)
                self._chk(lib.kwy_is_voiced_dev(h, _p(s.f0_pad), _p(s.ap_pad), s.Tp, K, fs, _p(s.voiced)))
                self._chk(lib.kwy_sp2mc_dev(h, _p(s.sp_pad), s.Tp, K, order, self.alpha, _p(s.mc_pad)))
                # make_feature(vuv='voiced', power='binalize', power_pivot='max')
                self._chk(lib.kwy_align_features_dev(h, _p(s.mc_pad), s.Tp, order + 1, _p(s.voiced), POWER_WEIGHT,
                                                     POWER_THRESHOLD, VUV_WEIGHT, _p(s.feat)))
            a, b = self.src, self.tgt
            cap = a.Tp + b.Tp + 2
            self.cap = cap
            self.path = torch.zeros((cap, 2), **i32)
            self.path_len = torch.zeros(1, dtype=torch.int64, device=self.dev)
            self.dist = torch.zeros(1, **f64)
            self.idx_x, self.idx_y = torch.zeros(cap, **i32), torch.zeros(cap, **i32)
            self.n_sel = torch.zeros(1, dtype=torch.int64, device=self.dev)
            self._chk(lib.kwy_fastdtw_dev(h, _p(a.feat), a.Tp, _p(b.feat), b.Tp, order + 2, self.radius, _p(self.dist),
                                          _p(self.path), _p(self.path_len)))
            # dtw_feature(strict=True) + align_even's cut
            self._chk(lib.kwy_align_even_dev(h, _p(self.path), _p(self.path_len), _p(a.feat), _p(b.feat), order + 2,
                                             1, 1, 1, a.Tp, b.Tp, P, _p(self.idx_x), _p(self.idx_y), cap,
                                             _p(self.n_sel)))
            halves = []
            for s, idx in ((a, self.idx_x), (b, self.idx_y)):
                mc_sel = torch.empty((cap, order + 1), **f64)
                self._chk(lib.kwy_gather_rows_dev(h, _p(s.mc_pad), s.Tp, order + 1, _p(idx), cap, _p(mc_sel)))
                static = mc_sel[:, 1:].contiguous()                        # drop the power coefficient
                delta = torch.empty((cap, 3 * order), **f64)
                self._chk(lib.kwy_delta_features_dev(h, _p(static), _p(self.n_sel), cap, order, _p(delta)))
                halves.append(delta)
            self.joint = torch.empty((cap, 6 * order), **f64)
            self._chk(lib.kwy_joint_rows_dev(h, _p(halves[0]), _p(halves[1]), _p(self.n_sel), cap, 3 * order, TRIM_EPS,
                                             _p(self.joint), _p(self.n_rows)))
            self._halves = halves        # alive until the stream is done with them

    def rows(self):
        with torch.cuda.stream(self.stream):
            n = int(self.n_rows.item())
            return self.joint[:n]


class _Lockstep:
    """Two streams with a library context each, shared by all waves of a lockstep driver (the default four hardware
    queues of the HIP runtime are enough: no GPU_MAX_HW_QUEUES)."""

    def __init__(self, device_index):
        self.dev = torch.device('cuda', device_index)
        self.main, self.side = torch.cuda.Stream(device=self.dev), torch.cuda.Stream(device=self.dev)
        self.ctx = _lib.Context(device_index, stream=self.main.cuda_stream)
        self.side_ctx = _lib.Context(device_index, stream=self.side.cuda_stream)

    def sync(self):
        self.main.synchronize()
        self.side.synchronize()


def _resident(side, dev, streams):
    """(x, f0, t) as device tensors that the given streams may use after the caller has dropped them"""
    out = []
    for a in side:
        t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        for s_ in streams:
            t.record_stream(s_)
        out.append(t)
    return out


class TrainWave:
    """<= 16 parallel pairs -> their rows of the training matrix, in lockstep through the batched entries of
    include/kwy.h (what TrainPair does pair by pair on a stream each):

        w.analyse()   enqueue CheapTrick + D4C of all utterances and their trim lengths; the lengths start their way
                      to the host (ONE read-back per wave)
        w.finish(X, cursor, pads)   (waits for the lengths) enqueue padding, voicing, sp2mc, DTW features, FastDTW,
                      strict filter + cut, deltas and the append of the joint rows behind the device-side cursor
    """

    def __init__(self, ls, fs, pairs, order=24, radius=32, frame_period=5.0):
        self.ls, self.fs, self.order, self.radius = ls, int(fs), int(order), int(radius)
        dev = ls.dev
        self.fft = lib.kwy_cheaptrick_fft_size(self.fs, 71.0)
        self.K = K = self.fft // 2 + 1
        from .backend import sptk
        self.alpha = sptk.mcepalpha(self.fs)
        self.n = len(pairs)
        f64 = dict(dtype=torch.float64, device=dev)
        P = PAD_LEN
        with torch.cuda.stream(ls.main):
            sides = [_resident(s, dev, (ls.main, ls.side)) for pair in pairs for s in pair]
            self.x, self.f0, self.t = ([s[k] for s in sides] for k in range(3))
            self.N = [len(v) for v in self.x]
            self.T = [len(v) for v in self.f0]
            off = np.concatenate(([0], np.cumsum([t + 2 * P for t in self.T]))).astype(np.int64)
            self.off = off
            rows = int(off[-1])
            self.rows = rows
            self.sp_pad = torch.empty((rows, K), **f64)
            self.ap_pad = torch.full((rows, K), 1 - SAFE_GUARD_MINIMUM, **f64)
            self.f0_pad = torch.empty(rows, **f64)
            self.voiced = torch.empty(rows, **f64)
            self.mc_pad = torch.empty((rows, order + 1), **f64)
            self.feat = torch.empty((rows, order + 2), **f64)
            ns = len(sides)
            self.keep_dev = torch.zeros(ns, dtype=torch.int64, device=dev)
            self.keep_host = torch.zeros(ns, dtype=torch.int64).pin_memory()
            reg = lambda a, i: a[int(off[i]):int(off[i + 1])]  # noqa: E731
            self.sp = [reg(self.sp_pad, i)[P:P + self.T[i]] for i in range(ns)]
            self.ap = [reg(self.ap_pad, i)[P:P + self.T[i]] for i in range(ns)]
            self.reg = reg
            self.j_env = _lib.utterance_array([(self.x[i], self.t[i], self.f0[i], self.sp[i]) for i in range(ns)])
            self.j_ap = _lib.utterance_array([(self.x[i], self.t[i], self.f0[i], self.ap[i]) for i in range(ns)])
            self.j_trim = _lib.job_array(_lib.TrimJob, [(self.sp[i], self.T[i], self.keep_dev[i:i + 1]) for i in range(ns)])
            cap = [self.T[2 * k] + self.T[2 * k + 1] + 4 * P + 2 for k in range(self.n)]
            self.path = [torch.zeros((c, 2), dtype=torch.int32, device=dev) for c in cap]
            self.path_len = torch.zeros(self.n, dtype=torch.int64, device=dev)
            self.dist = torch.zeros(self.n, **f64)
            self.n_rows = torch.zeros(self.n, dtype=torch.int64, device=dev)
        self.frames = sum(self.T[0::2])

    def analyse(self):
        ls, fs, fft, K, ns = self.ls, self.fs, self.fft, self.K, 2 * self.n
        ls.side.wait_stream(ls.main)                # (the buffers were made on the main stream)
        with torch.cuda.stream(ls.side):
            _lib.check(ls.side_ctx, lib.kwy_d4c_batch_dev(ls.side_ctx.handle, self.j_ap, ns, fs, 0.85, fft))
            self.ap_done = torch.cuda.Event()
            self.ap_done.record(ls.side)
        with torch.cuda.stream(ls.main):
            h = ls.ctx.handle
            _lib.check(ls.ctx, lib.kwy_cheaptrick_batch_dev(h, self.j_env, ns, fs, -0.15, 71.0, fft, float(fs)))
            _lib.check(ls.ctx, lib.kwy_trim_length_batch_dev(h, self.j_trim, ns, K, TRIM_EPS))
            self.keep_host.copy_(self.keep_dev, non_blocking=True)
            self.keep_ready = torch.cuda.Event()
            self.keep_ready.record(ls.main)

    def finish(self, X, cursor, pads):
        """pads(rows): fills the wave's pad rows -- a list of 4 n device views (source head, source tail, target
        head, target tail, pair after pair) -- on the main stream"""
        ls, fs, K, order, P, ns = self.ls, self.fs, self.K, self.order, PAD_LEN, 2 * self.n
        self.keep_ready.synchronize()
        keep = [int(v) for v in self.keep_host.tolist()]
        Tp = [k + 2 * P for k in keep]
        reg = self.reg
        with torch.cuda.stream(ls.main):
            h = ls.ctx.handle
            chk = lambda rc: _lib.check(ls.ctx, rc)  # noqa: E731
            sp_r = [reg(self.sp_pad, i) for i in range(ns)]
            pads([blk for i in range(ns) for blk in (sp_r[i][:P], sp_r[i][P + keep[i]:Tp[i]])])
            ls.main.wait_event(self.ap_done)
            J = _lib.job_array
            chk(lib.kwy_train_pad_batch_dev(h, J(_lib.PadJob, [(self.f0[i], keep[i], reg(self.f0_pad, i), reg(self.ap_pad, i),
                                                               reg(self.voiced, i)) for i in range(ns)]), ns, K, fs, P))
            chk(lib.kwy_sp2mc_dev(h, _p(self.sp_pad), self.rows, K, order, self.alpha, _p(self.mc_pad)))
            # make_feature(vuv='voiced', power='binalize', power_pivot='max')
            chk(lib.kwy_align_features_batch_dev(h, J(_lib.AlignJob, [(reg(self.mc_pad, i), reg(self.voiced, i), Tp[i],
                                                                       reg(self.feat, i)) for i in range(ns)]),
                                                 ns, order + 1, POWER_WEIGHT, POWER_THRESHOLD, VUV_WEIGHT))
            chk(lib.kwy_fastdtw_batch_dev(h, J(_lib.DtwJob, [(reg(self.feat, 2 * k), Tp[2 * k], reg(self.feat, 2 * k + 1),
                                                              Tp[2 * k + 1], self.dist[k:k + 1], self.path[k],
                                                              self.path_len[k:k + 1]) for k in range(self.n)]),
                                          self.n, order + 2, self.radius))
            # dtw_feature(strict=True) + align_even's cut, deltas, hstack + remove_zeros_frames, append
            chk(lib.kwy_train_rows_batch_dev(
                h, J(_lib.TrainJob, [(self.path[k], self.path_len[k:k + 1], reg(self.feat, 2 * k), reg(self.feat, 2 * k + 1),
                                      reg(self.mc_pad, 2 * k), reg(self.mc_pad, 2 * k + 1), Tp[2 * k], Tp[2 * k + 1],
                                      self.n_rows[k:k + 1]) for k in range(self.n)]),
                self.n, order, 1, 1, 1, P, TRIM_EPS, _p(X), X.shape[0], _p(cursor)))
        self.keep = keep


class ConvertPipeline(_Graphed):
    """convert_voice.convert(diffvc=False) of one utterance, HBM-resident: analyse -> mel-cepstrum -> GMM/MLPG
    conversion (c0 kept) -> spectrum -> synthesis with the utterance's own f0 and aperiodicity.

    mcep_fs: the sampling rate the converter was trained at, when it differs from the utterance's
    (`--mcep-fs`; MelCepstrumFeatureConverter.convert and the `feature.mel_cepstrum = ...` that follows it,
    /root/reference/kwiiyatta/converter/mcep.py:47-61, vocoder/mcep.py:31-45, vocoder/abc/feature.py): the
    mel-cepstrum travels to the converter's rate and back through its spectrum -- mc2sp on the vocoder's grid of
    the old rate, the bins cut at the new Nyquist frequency or extended by "silent" bins |N(0, EPS / old_fs)|, sp2mc
    with the new rate's alpha (any bin count: the dense form).  The silent bins are one numpy draw of (T, missing)
    values per call; `rng` (a backend.nprandom.DeviceRandomState) draws them on the device in the reference's
    order, so a seeded run equals the Python API path."""

    def __init__(self, device_index, fs, utterance, gmm, order=24, frame_period=5.0, stream=None, ctx=None,
                 mcep_fs=None, rng=None):
        self.dev = torch.device('cuda', device_index)
        self.fs, self.order, self.frame_period = int(fs), int(order), float(frame_period)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.dev)
        self.ctx = ctx if ctx is not None else _lib.Context(device_index, stream=self.stream.cuda_stream)
        self.fft = lib.kwy_cheaptrick_fft_size(self.fs, 71.0)
        self.K = self.fft // 2 + 1
        from .backend import sptk
        self.alpha = sptk.mcepalpha(self.fs)
        self.gmm = gmm
        assert gmm.D2 == 6 * order
        with torch.cuda.stream(self.stream):
            self.gmm_model = gmm.model(diff=False)
        x, f0, t = utterance
        self.N, self.T = len(x), len(f0)
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.mcep_fs = int(mcep_fs) if mcep_fs is not None and int(mcep_fs) != self.fs else None
        self.rng = rng
        with torch.cuda.stream(self.stream):
            for a in (x, f0, t):
                if torch.is_tensor(a):
                    a.record_stream(self.stream)        # (cloned below on this stream; the caller may drop it)
            self.x, self.f0, self.t = (a.clone() if torch.is_tensor(a) else
                                       torch.from_numpy(np.ascontiguousarray(a)).to(self.dev) for a in (x, f0, t))
            self.sp = torch.empty((self.T, self.K), **f64)
            self.ap = torch.empty((self.T, self.K), **f64)
            self.mc = torch.empty((self.T, order + 1), **f64)
            self.mc_x = torch.empty((self.T, order), **f64)
            self.mc_y = torch.empty((self.T, order), **f64)
            self.mc_conv = torch.empty((self.T, order + 1), **f64)
            self.sp_conv = torch.empty((self.T, self.K), **f64)
            self.ylen = lib.kwy_synth_length(self.T, self.frame_period, self.fs)
            self.wave = torch.empty(self.ylen, **f64)
            if self.mcep_fs is not None:
                if rng is None:
                    raise ValueError('ConvertPipeline(mcep_fs=...) needs rng: the silent bins of the wider spectrum '
                                     'are random draws')
                fc = self.mcep_fs
                self.alpha_c = sptk.mcepalpha(fc)
                self.Kc = lib.kwy_cheaptrick_fft_size(fc, 71.0) // 2 + 1         # the vocoder's grid at the other rate
                self.n_there = self.K * fc // self.fs                            # our grid carried to the converter's rate
                self.n_back = self.Kc * self.fs // fc                            # its grid carried back to ours
                self.spec_u = torch.empty((self.T, self.K), **f64)
                self.there = torch.empty((self.T, self.n_there), **f64)
                self.mc_c = torch.empty((self.T, order + 1), **f64)
                self.mc_c2 = torch.empty((self.T, order + 1), **f64)
                self.spec_c = torch.empty((self.T, self.Kc), **f64)
                self.back = torch.empty((self.T, self.n_back), **f64)
                # the one "up" direction of the round trip appends random silent bins (drawn per run)
                miss = self.n_there - self.K if fc > self.fs else self.n_back - self.Kc
                self.silent = torch.empty((self.T, miss), **f64)
        self.stream.synchronize()
        self.frames = self.T

    def load(self, utterance):
        """another utterance of the same shape into the existing buffers (numpy arrays or device tensors;
        asynchronous on the pipeline's stream)"""
        x, f0, t = utterance
        if len(x) != self.N or len(f0) != self.T:
            raise ValueError('ConvertPipeline.load: shape differs from the pipeline\'s')
        with torch.cuda.stream(self.stream):
            for dst, src in ((self.x, x), (self.f0, f0), (self.t, t)):
                if torch.is_tensor(src):
                    src.record_stream(self.stream)      # (the caller may drop it while the copy is still queued)
                dst.copy_(src if torch.is_tensor(src) else torch.from_numpy(np.ascontiguousarray(src)),
                          non_blocking=True)

    def _convert_across_rates(self, chk, h):
        """self.mc (utterance rate) -> self.mc_conv (utterance rate) through the converter's rate"""
        T, order, fs, fc = self.T, self.order, self.fs, self.mcep_fs
        EPS = 2.220446049250313e-16
        up_first = fc > fs
        if True:                    # (one block: the order below IS the order of the reference's calls)
            # MelCepstrum.resample_data(fc): spectrum on our grid, cut / extended, coefficients with the new alpha
            chk(lib.kwy_mc2sp_dev(h, _p(self.mc), T, order, self.alpha, self.fft, _p(self.spec_u)))
            if up_first:
                self.rng.stream.wait_stream(self.stream)
                self.rng.abs_normal(EPS / fs, out=self.silent)
                self.stream.wait_event(self.rng.record_event())
                self.there[:, :self.K].copy_(self.spec_u)
                self.there[:, self.K:].copy_(self.silent)
            else:
                self.there.copy_(self.spec_u[:, :self.n_there])
            chk(lib.kwy_sp2mc_dev(h, _p(self.there), T, self.n_there, order, self.alpha_c, _p(self.mc_c)))
            # the conversion itself, at the converter's rate (c0 kept)
            chk(lib.kwy_convert_mcep_dev(h, _p(self.mc_c), T, order, self.gmm.M, _p(self.gmm_model), _p(self.mc_c2)))
            # `feature.mel_cepstrum = converted` -> resample_data(fs): back through the converter rate's grid
            chk(lib.kwy_mc2sp_dev(h, _p(self.mc_c2), T, order, self.alpha_c, 2 * (self.Kc - 1), _p(self.spec_c)))
            if not up_first:
                self.rng.stream.wait_stream(self.stream)
                self.rng.abs_normal(EPS / fc, out=self.silent)
                self.stream.wait_event(self.rng.record_event())
                self.back[:, :self.Kc].copy_(self.spec_c)
                self.back[:, self.Kc:].copy_(self.silent)
            else:
                self.back.copy_(self.spec_c[:, :self.n_back])
            chk(lib.kwy_sp2mc_dev(h, _p(self.back), T, self.n_back, order, self.alpha, _p(self.mc_conv)))

    def run(self):
        h, fs, fft, K, order, T = self.ctx.handle, self.fs, self.fft, self.K, self.order, self.T
        chk = lambda rc: _lib.check(self.ctx, rc)  # noqa: E731
        with torch.cuda.stream(self.stream):
            chk(lib.kwy_cheaptrick_dev(h, _p(self.x), self.N, fs, _p(self.t), _p(self.f0), T, -0.15, 71.0, fft,
                                       float(fs), _p(self.sp)))
            chk(lib.kwy_d4c_dev(h, _p(self.x), self.N, fs, _p(self.t), _p(self.f0), T, 0.85, fft, _p(self.ap)))
            chk(lib.kwy_sp2mc_dev(h, _p(self.sp), T, K, order, self.alpha, _p(self.mc)))
            if self.mcep_fs is None:
                chk(lib.kwy_convert_mcep_dev(h, _p(self.mc), T, order, self.gmm.M, _p(self.gmm_model), _p(self.mc_conv)))
            else:
                self._convert_across_rates(chk, h)
            chk(lib.kwy_mc2sp_dev(h, _p(self.mc_conv), T, order, self.alpha, fft, _p(self.sp_conv)))
            chk(lib.kwy_synthesize_dev(h, _p(self.f0), T, _p(self.sp_conv), _p(self.ap), fft, self.frame_period, fs,
                                       float(fs), self.ylen, _p(self.wave)))

    def sync(self):
        self.ctx.sync()

    def contexts(self):
        """(a captured pass also holds addresses of the generator's scratch arena)"""
        cs = [self.ctx]
        if self.mcep_fs is not None and self.rng is not None and self.rng.ctx is not self.ctx:
            cs.append(self.rng.ctx)
        return cs


class ConvertWave:
    """<= 16 utterances analysed and rendered in lockstep (the batched entries of include/kwy.h on two streams): with a
    prepared GMM model the mel-cepstra are converted in between (convert_voice.convert(diffvc=False) of every file,
    /root/reference/kwiiyatta/convert_voice.py:35-46), without one the features are resynthesised as they are
    (resynthesize_voice.py:46-79, BASELINE config 4).  Utterances of any lengths; `wave[i]` are views of one block.

    An utterance is an (x, f0, t) triple, or a bare waveform (numpy array / device tensor): then its f0 track is
    extracted inside the wave -- DIO + StoneMask on the device (kwy_dio_batch_dev, kwy_stonemask_batch_dev:
    Analyzer.extract_f0, /root/reference/kwiiyatta/vocoder/world.py:33-41) -- and `f0_status` holds one word per
    utterance to read back (non-zero: DIO's zero-crossing buffer overflowed).  pcm=True: the post-step and the 16-bit
    samples on the device as well (kwy_finish_pcm16_batch_dev: vocoder/abc/synthesizer.py:11-20, wavfile.py:8-29):
    `pcm[i]` int16 views, 2 bytes per sample to download.  diff=True (with a GMM): also the DIFFERENTIAL output of
    every file -- the input waveform through the MLSA filter of the differential conversion, convert(diffvc=True),
    /root/reference/kwiiyatta/convert_voice.py:19,39-40, filter/mlsa.py:9-30 -- as `wave_diff[i]` (and `pcm_diff[i]`:
    Wavdata.save's normalisation only, a filtered waveform has no synthesis post-step)."""

    def __init__(self, ls, fs, utterances, gmm=None, order=24, frame_period=5.0, pcm=False, diff=False, defer_mlsa=False):
        self.ls, self.fs, self.order, self.frame_period = ls, int(fs), int(order), float(frame_period)
        self.diff = bool(diff) and gmm is not None
        self.defer_mlsa = bool(defer_mlsa)       # the caller launches the MLSA recursions of several waves together
        self.wav_in = len(utterances) > 0 and not isinstance(utterances[0], (tuple, list))
        dev = ls.dev
        self.fft = lib.kwy_cheaptrick_fft_size(self.fs, 71.0)
        self.K = K = self.fft // 2 + 1
        from .backend import sptk
        self.alpha = sptk.mcepalpha(self.fs)
        self.gmm = gmm
        self.n = n = len(utterances)
        f64 = dict(dtype=torch.float64, device=dev)
        with torch.cuda.stream(ls.main):
            self.model = gmm.model(diff=False) if gmm is not None else None
            cut = lambda a, o, i: a[int(o[i]):int(o[i + 1])]  # noqa: E731
            if self.wav_in:
                self.x = [_resident((u,), dev, (ls.main, ls.side))[0] for u in utterances]
                self.T = [int(lib.kwy_dio_frames(self.fs, v.numel(), self.frame_period)) for v in self.x]
                off = np.concatenate(([0], np.cumsum(self.T))).astype(np.int64)
                blocks = [torch.empty(int(off[-1]), **f64) for _ in range(3)]
                self.t, self.f0_dio, self.f0 = ([cut(b, off, i) for i in range(n)] for b in blocks)
                self.f0_status = torch.zeros(n, dtype=torch.int32, device=dev)
                self.j_dio = _lib.job_array(_lib.F0Job, [(self.x[i], self.x[i].numel(), self.t[i], self.f0_dio[i],
                                                          self.f0_status[i:i + 1]) for i in range(n)])
                self.j_sm = _lib.utterance_array([(self.x[i], self.t[i], self.f0_dio[i], self.f0[i]) for i in range(n)])
            else:
                us = [_resident(u, dev, (ls.main, ls.side)) for u in utterances]
                self.x, self.f0, self.t = ([u[k] for u in us] for k in range(3))
                self.T = [len(v) for v in self.f0]
                off = np.concatenate(([0], np.cumsum(self.T))).astype(np.int64)
                self.f0_status = None
            self.rows = int(off[-1])
            # (with a GMM nothing reads the analysed envelopes but sp2mc: CheapTrick hands over mel-cepstra instead,
            # kwy_cheaptrick_mcep_batch_dev)
            self.sp_all = torch.empty((self.rows, K), **f64) if gmm is None else None
            self.ap_all = torch.empty((self.rows, K), **f64)
            self.ylen = [int(lib.kwy_synth_length(t, self.frame_period, self.fs)) for t in self.T]
            yo = np.concatenate(([0], np.cumsum(self.ylen))).astype(np.int64)
            self.wave_all = torch.empty(int(yo[-1]), **f64)
            self.wave = [cut(self.wave_all, yo, i) for i in range(n)]
            self.pcm = None
            if pcm:
                self.pcm_all = torch.zeros(int(yo[-1]), dtype=torch.int16, device=dev)
                self.pcm = [cut(self.pcm_all, yo, i) for i in range(n)]
                self.j_fin = _lib.job_array(_lib.FinishJob, [(self.wave[i], self.ylen[i], self.T[i], self.pcm[i])
                                                             for i in range(n)])
            self.plan = [torch.empty(int(lib.kwy_synth_plan_bytes(y)), dtype=torch.uint8, device=dev) for y in self.ylen]
            if gmm is not None:
                assert gmm.D2 == 6 * order
                self.mc = torch.empty((self.rows, order + 1), **f64)
            sp = [cut(self.sp_all if gmm is None else self.mc, off, i) for i in range(n)]
            ap = [cut(self.ap_all, off, i) for i in range(n)]
            self.j_env = _lib.utterance_array([(self.x[i], self.t[i], self.f0[i], sp[i]) for i in range(n)])
            self.j_ap = _lib.utterance_array([(self.x[i], self.t[i], self.f0[i], ap[i]) for i in range(n)])
            self.j_plan = _lib.job_array(_lib.SynthPlanJob, [(self.f0[i], self.T[i], self.ylen[i], self.plan[i]) for i in range(n)])
            if gmm is not None:
                self.mc_conv = torch.empty((self.rows, order + 1), **f64)
                self.sp_conv = torch.empty((self.rows, K), **f64)
                self.j_conv = _lib.job_array(_lib.ConvertJob, [(cut(self.mc, off, i), self.T[i], cut(self.mc_conv, off, i))
                                                               for i in range(n)])
                sp = [cut(self.sp_conv, off, i) for i in range(n)]
            if self.diff:
                self.model_diff = gmm.model(diff=True)
                self.mc_diff = torch.empty((self.rows, order + 1), **f64)
                self.j_conv_diff = _lib.job_array(_lib.ConvertJob, [(cut(self.mc, off, i), self.T[i], cut(self.mc_diff, off, i))
                                                                    for i in range(n)])
                xo = np.concatenate(([0], np.cumsum([v.numel() for v in self.x]))).astype(np.int64)
                self.wave_diff_all = torch.empty(int(xo[-1]), **f64)
                self.wave_diff = [cut(self.wave_diff_all, xo, i) for i in range(n)]
                self.mc_diff_rows = [cut(self.mc_diff, off, i) for i in range(n)]
                self.j_mlsa = _lib.job_array(_lib.MlsaJob, [(self.x[i], self.x[i].numel(), self.mc_diff_rows[i], self.T[i],
                                                            self.wave_diff[i]) for i in range(n)])
                self.hop = int(self.fs * (self.frame_period * 0.001))
                self.pcm_diff = None
                if pcm:
                    self.pcm_diff_all = torch.zeros(int(xo[-1]), dtype=torch.int16, device=dev)
                    self.pcm_diff = [cut(self.pcm_diff_all, xo, i) for i in range(n)]
                    self.j_fin_diff = _lib.job_array(_lib.FinishJob, [(self.wave_diff[i], self.x[i].numel(), 0, self.pcm_diff[i])
                                                                      for i in range(n)])
            self.j_render = _lib.synth_job_array([(self.plan[i], sp[i], ap[i], self.wave[i]) for i in range(n)])
        self.frames = int(sum(self.T))

    def run(self):
        ls, fs, fft, K, order, n = self.ls, self.fs, self.fft, self.K, self.order, self.n
        if self.wav_in:
            with torch.cuda.stream(ls.main):
                _lib.check(ls.ctx, lib.kwy_dio_batch_dev(ls.ctx.handle, self.j_dio, n, fs, 71.0, 800.0, 2.0,
                                                         self.frame_period, 1, 0.1))
                _lib.check(ls.ctx, lib.kwy_stonemask_batch_dev(ls.ctx.handle, self.j_sm, n, fs))
        ls.side.wait_stream(ls.main)
        with torch.cuda.stream(ls.side):
            hs = ls.side_ctx.handle
            _lib.check(ls.side_ctx, lib.kwy_d4c_batch_dev(hs, self.j_ap, n, fs, 0.85, fft))
            _lib.check(ls.side_ctx, lib.kwy_synth_plan_batch_dev(hs, self.j_plan, n, fft, self.frame_period, fs))
        with torch.cuda.stream(ls.main):
            h = ls.ctx.handle
            chk = lambda rc: _lib.check(ls.ctx, rc)  # noqa: E731
            if self.gmm is None:
                chk(lib.kwy_cheaptrick_batch_dev(h, self.j_env, n, fs, -0.15, 71.0, fft, float(fs)))
            else:
                chk(lib.kwy_cheaptrick_mcep_batch_dev(h, self.j_env, n, fs, -0.15, 71.0, fft, float(fs), order, self.alpha))
                chk(lib.kwy_convert_mcep_batch_dev(h, self.j_conv, n, order, self.gmm.M, _p(self.model)))
                chk(lib.kwy_mc2sp_dev(h, _p(self.mc_conv), self.rows, order, self.alpha, fft, _p(self.sp_conv)))
            ls.main.wait_stream(ls.side)
            chk(lib.kwy_synth_render_batch_dev(h, self.j_render, n, fft, self.frame_period, fs, float(fs)))
            from .pipeline import PIECE_CEILING
            if self.pcm is not None:
                chk(lib.kwy_finish_pcm16_batch_dev(h, self.j_fin, n, fs, 1, PIECE_CEILING, 1, PIECE_CEILING))
            if self.diff:
                # the differential conversion of the same mel-cepstra, its filter over the INPUT waveforms: all
                # utterances' recursions side by side (one wavefront each)
                chk(lib.kwy_convert_mcep_batch_dev(h, self.j_conv_diff, n, order, self.gmm.M, _p(self.model_diff)))
                if not self.defer_mlsa:
                    self.run_mlsa(ls.ctx)

    def mlsa_rows(self):
        """the wave's MLSA jobs as rows (for ONE launch over the utterances of several waves: a recursion occupies one
        wavefront for 0.6 us per sample whatever else runs, so all files of a batch should recurse side by side)"""
        return [(self.x[i], self.x[i].numel(), self.mc_diff_rows[i], self.T[i], self.wave_diff[i]) for i in range(self.n)]

    def run_mlsa(self, ctx):
        from .pipeline import PIECE_CEILING
        chk = lambda rc: _lib.check(ctx, rc)  # noqa: E731
        chk(lib.kwy_mlsa_filter_batch_dev(ctx.handle, self.j_mlsa, self.n, self.order, self.alpha, 4, self.hop, 1))
        self.finish_diff(ctx)

    def finish_diff(self, ctx):
        from .pipeline import PIECE_CEILING
        if self.pcm_diff is not None:
            _lib.check(ctx, lib.kwy_finish_pcm16_batch_dev(ctx.handle, self.j_fin_diff, self.n, self.fs, 0, PIECE_CEILING, 1,
                                                           PIECE_CEILING))


def _lockstep_batch(utterances, fs, device_index, gmm, order, frame_period, ls, keep, wave_size=16, pcm=False, diff=False):
    """utterances in waves of `wave_size` through ConvertWave; keep(i, waveform view[, pcm view]) on the main stream.
    Bare waveforms get their f0 on the device; the DIO status words of all waves are read back ONCE at the end."""
    ls = ls if ls is not None else _Lockstep(device_index)
    held, status, waves_diff = [], [], []
    for w0 in range(0, len(utterances), wave_size):
        wv = ConvertWave(ls, fs, utterances[w0:w0 + wave_size], gmm=gmm, order=order, frame_period=frame_period, pcm=pcm,
                         diff=diff, defer_mlsa=diff)
        wv.run()
        with torch.cuda.stream(ls.main):
            for i in range(wv.n):
                if diff:
                    keep(w0 + i, wv.wave[i], wv.pcm[i] if pcm else None, wv.wave_diff[i], wv.pcm_diff[i] if pcm else None)
                elif pcm:
                    keep(w0 + i, wv.wave[i], wv.pcm[i])
                else:
                    keep(w0 + i, wv.wave[i])
        if wv.f0_status is not None:
            status.append(wv.f0_status)
        if diff:
            waves_diff.append(wv)            # (kept: its inputs and mel-cepstra feed the filter launch below)
        held.append(wv)
        while len(held) > 2:
            held.pop(0)
    if waves_diff:
        # the differential outputs of ALL files: one pass of launches over every recursion (64 per launch; two contexts
        # alternate so that consecutive launches overlap), then their post-step
        rows = [r for wv in waves_diff for r in wv.mlsa_rows()]
        wv0 = waves_diff[0]
        ls.side.wait_stream(ls.main)
        for k, c0 in enumerate(range(0, len(rows), 64)):
            ctx = (ls.ctx, ls.side_ctx)[k % 2]
            with torch.cuda.stream((ls.main, ls.side)[k % 2]):
                chunk = rows[c0:c0 + 64]
                _lib.check(ctx, lib.kwy_mlsa_filter_batch_dev(ctx.handle, _lib.job_array(_lib.MlsaJob, chunk), len(chunk), order,
                                                              wv0.alpha, 4, wv0.hop, 1))
        ls.main.wait_stream(ls.side)
        with torch.cuda.stream(ls.main):
            for wv in waves_diff:
                wv.finish_diff(ls.ctx)
    ls.sync()
    if status and bool(torch.cat(status).any().item()):
        bad = torch.nonzero(torch.cat(status)).flatten().tolist()
        raise RuntimeError(f'dio: zero-crossing buffer overflow in utterance(s) {bad} (signal too noisy for the band filters)')
    return ls


class StreamPool:
    """`n` HIP streams with one library context each (a context owns constant tables and a scratch arena: built
    once per stream, not once per utterance).

    Streams only overlap when they land on different hardware queues; the HIP runtime creates 4 unless the
    application exports GPU_MAX_HW_QUEUES before HIP starts (bench.py and bench_corpus.py set 64: more than they
    have streams, see bench.py).  The package does
    not touch the environment: with fewer queues than streams the pool still works, the streams just share."""

    def __init__(self, device_index, n):
        self.dev = torch.device('cuda', device_index)
        import os
        import warnings
        queues = os.environ.get('GPU_MAX_HW_QUEUES')
        if n > 4 and queues is None:
            warnings.warn(f'{n} streams, but GPU_MAX_HW_QUEUES is not set: the HIP runtime maps all streams onto 4 '
                          f'hardware queues and most of the overlap between utterances is lost (1.43 M instead of '
                          f'1.91 M frames/s in bench.py); export it before the first HIP call', RuntimeWarning,
                          stacklevel=2)
        self.streams = [torch.cuda.Stream(device=self.dev) for _ in range(max(1, n))]
        self.contexts = [_lib.Context(device_index, stream=s.cuda_stream) for s in self.streams]

    def __len__(self):
        return len(self.streams)


def shard_block(n_items, rank, world_size):
    """Contiguous block of items for `rank`: the concatenation over ranks is the original order."""
    if not (0 <= rank < world_size):
        raise ValueError(f'rank {rank} outside world of size {world_size}')
    lo = n_items * rank // world_size
    hi = n_items * (rank + 1) // world_size
    return list(range(lo, hi))


class _silence_ahead:
    """The pad spectra of `n_pairs` pairs, drawn by a helper thread in pair order from numpy's global legacy generator
    (the reference's source, `draw_silence`) while the caller enqueues GPU work: the generator is serial (4.6 ms per
    pair at 48 kHz) and numpy releases the GIL inside it.  The caller must not use `np.random` between construction
    and `stop()`; `stop()` (always called by build_training_matrix, also when a pair raises) ends the thread, so a
    failed run does not leave a thread behind that keeps consuming global draws."""

    def __init__(self, n_pairs, fs, depth):
        import queue
        import threading
        K = lib.kwy_cheaptrick_fft_size(int(fs), 71.0) // 2 + 1
        self.q = queue.Queue(maxsize=max(1, depth))
        self.halt = threading.Event()

        def work():
            for _ in range(n_pairs):
                item = [draw_silence(fs, K) for _ in range(4)]
                while not self.halt.is_set():
                    try:
                        self.q.put(item, timeout=0.05)
                        break
                    except queue.Full:
                        continue
                if self.halt.is_set():
                    return
        self.thread = threading.Thread(target=work, daemon=True)
        self.thread.start()

    def get(self):
        return self.q.get()

    def stop(self):
        self.halt.set()
        self.thread.join(timeout=5.0)


class _upload_ahead:
    """The waveforms, f0 tracks and frame times of the pairs as device tensors, uploaded by a helper thread in pair
    order while the caller enqueues GPU work: a copy from pageable host memory blocks its caller (0.2 ms per 5 s
    waveform at 48 kHz, six copies per pair: 0.5 ms of the ~1.3 ms of host time a pair costs) and releases the
    interpreter lock meanwhile.  A copy is complete when the helper hands the tensors over.  `stop()` ends the thread."""

    def __init__(self, pairs, dev, depth):
        import queue
        import threading
        self.q = queue.Queue(maxsize=max(1, depth))
        self.halt = threading.Event()

        def side_up(side):
            return tuple(a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in side)

        def hand(item):
            while not self.halt.is_set():
                try:
                    self.q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    continue
            return False

        def work():
            try:
                for it in pairs:
                    if not hand(tuple(side_up(side) for side in it)):
                        return
            except Exception as exc:        # handed to the consumer: it raises where the pair would have been built
                hand(exc)
        self.thread = threading.Thread(target=work, daemon=True)
        self.thread.start()

    def get(self):
        item = self.q.get()
        if isinstance(item, Exception):
            raise item
        return item

    def stop(self):
        self.halt.set()
        self.thread.join(timeout=5.0)


def build_training_matrix(pairs, fs, device_index=0, order=24, radius=32, frame_period=5.0, streams=16,
                          silence_for=None, pool=None, rng=None, pairs_before=0, driver=None, lockstep=None,
                          wave_pairs=16):
    """driver='lockstep' (default): waves of `wave_pairs` pairs through the batched entries on two streams
    (`TrainWave`; `lockstep`: a _Lockstep to reuse), rows appended behind a device-side cursor, one read-back per wave;
    driver='streams': round 3's pair-per-stream driver (`TrainPair`, below).  Same matrix either way."""
    driver = driver or ('streams' if pool is not None else 'lockstep')
    if driver == 'lockstep':
        return _build_training_matrix_lockstep(pairs, fs, device_index, order, radius, frame_period, silence_for, rng,
                                               pairs_before, lockstep, wave_pairs)
    return _build_training_matrix_streams(pairs, fs, device_index, order, radius, frame_period, streams, silence_for,
                                          pool, rng, pairs_before)


def _build_training_matrix_lockstep(pairs, fs, device_index, order, radius, frame_period, silence_for, rng, pairs_before,
                                    ls, wave_pairs):
    dev = torch.device('cuda', device_index)
    ls = ls if ls is not None else _Lockstep(device_index)
    K = lib.kwy_cheaptrick_fft_size(int(fs), 71.0) // 2 + 1
    scale = 2.220446049250313e-16 / fs
    wave_pairs = max(1, min(16, int(wave_pairs)))
    if not pairs:
        return torch.empty((0, 6 * order), dtype=torch.float64, device=dev), 0
    if rng is not None and pairs_before:
        with torch.cuda.stream(ls.main):
            sink = [torch.empty((PAD_LEN, K), dtype=torch.float64, device=dev) for _ in range(4 * 16)]
            left = pairs_before
            while left > 0:
                take = min(16, left)
                rng.abs_normal_blocks(scale, sink[:4 * take], ctx=ls.ctx)
                left -= take
    ahead = _silence_ahead(len(pairs), fs, 2 * wave_pairs) if silence_for is None and rng is None else None
    uploads = _upload_ahead(pairs, dev, 3 * wave_pairs)
    # capacity of the matrix: a pair yields at most one row per path cell of its un-padded stretch
    cap_rows = sum(len(p[0][1]) + len(p[1][1]) for p in pairs)
    frames = 0
    done = [0]                # pairs whose pads have been handed out

    def pads(rows):
        n_pairs = len(rows) // 4
        if silence_for is not None or ahead is not None:
            for k in range(n_pairs):
                sil = silence_for(done[0] + k) if silence_for is not None else ahead.get()
                for dst, a in zip(rows[4 * k:4 * k + 4], sil):
                    dst.copy_(a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a)), non_blocking=True)
        else:
            rng.abs_normal_blocks(scale, rows, ctx=ls.ctx)
        done[0] += n_pairs

    try:
        with torch.cuda.stream(ls.main):
            X = torch.empty((cap_rows, 6 * order), dtype=torch.float64, device=dev)
            cursor = torch.zeros(1, dtype=torch.int64, device=dev)
        prev, held = None, []
        worst = torch.zeros(1, dtype=torch.int64, device=dev)      # min over all pairs' n_rows: < 0 = a pair was dropped

        def close(wave):
            wave.finish(X, cursor, pads)
            with torch.cuda.stream(ls.main):
                torch.minimum(worst, wave.n_rows.min().reshape(1), out=worst)
        for w0 in range(0, len(pairs), wave_pairs):
            chunk = [uploads.get() for _ in range(len(pairs[w0:w0 + wave_pairs]))]
            wave = TrainWave(ls, fs, chunk, order=order, radius=radius, frame_period=frame_period)
            wave.analyse()                     # enqueued BEFORE the host waits for the previous wave's lengths
            if prev is not None:
                close(prev)
                held.append(prev)
            frames += wave.frames
            prev = wave
            while len(held) > 2:
                held.pop(0)                    # (its buffers: all uses are ordered on the main stream before reuse)
        close(prev)
        with torch.cuda.stream(ls.main):
            n_rows, dropped = (int(v) for v in torch.cat((cursor, worst)).tolist())     # ONE read-back
        ls.sync()
        if dropped < 0:
            # k_tr_rows_place marks a pair that did not fit behind the cursor with -1 - rows and drops it; the capacity
            # above (one row per frame of both sides) bounds every pair's rows, so this is a bug, never a data property
            raise RuntimeError(f'training matrix: a pair of {-1 - dropped} rows did not fit the capacity of {cap_rows}')
    finally:
        if ahead is not None:
            ahead.stop()
        uploads.stop()
    torch.cuda.current_stream(dev).synchronize()
    # (a view would keep the whole capacity block alive: twice the rows actually used, or more)
    return (X[:n_rows].clone() if n_rows * 4 < cap_rows * 3 else X[:n_rows]), frames


def _build_training_matrix_streams(pairs, fs, device_index=0, order=24, radius=32, frame_period=5.0, streams=16,
                                   silence_for=None, pool=None, rng=None, pairs_before=0):
    """pairs: list of ((x, f0, t), (x, f0, t)) numpy triples of THIS rank, in corpus order.  Returns the
    (n, 2*3*order) float64 device tensor of make_dataset_to_array and the number of source frames analysed.
    Pairs are processed `streams` at a time, each on its own stream (`pool`: a StreamPool to use instead of a new one).
    The pad spectra of pair i come from
      silence_for(i)   if given (four host arrays), else from
      rng              a DeviceRandomState: drawn on the GPU, in the reference's order, from numpy's legacy stream;
                       `pairs_before` pairs (those of the ranks before this one) are drawn and discarded first, so
                       that every pair gets the pads it would get on one rank, whatever the number of ranks; else from
      np.random        the global generator on the host, as the reference draws them, one wave of pairs ahead of the
                       GPU on a helper thread (bit-equal to the Python API path under np.random.seed)."""
    dev = torch.device('cuda', device_index)
    if pool is None:
        pool = StreamPool(device_index, streams)
    K = lib.kwy_cheaptrick_fft_size(int(fs), 71.0) // 2 + 1
    scale = 2.220446049250313e-16 / fs
    if rng is not None and pairs_before:
        with torch.cuda.stream(rng.stream):
            sink = [torch.empty((PAD_LEN, K), dtype=torch.float64, device=dev) for _ in range(4)]
            for _ in range(pairs_before):
                rng.abs_normal_blocks(scale, sink)
    ahead = _silence_ahead(len(pairs), fs, 2 * len(pool)) if silence_for is None and rng is None and pairs else None
    uploads = _upload_ahead(pairs, dev, 2 * len(pool)) if pairs else None
    blocks, frames = [], 0

    def finish(wave):
        """the host-dependent half of a wave: trim lengths back, alignment and row extraction enqueued, rows collected"""
        nonlocal frames
        for p in wave:
            p.align()
        for p in wave:
            blocks.append(p.rows().clone())  # enqueued on the default stream after rows() has synchronised
            frames += p.frames
        torch.cuda.current_stream(dev).synchronize()     # the copies are done before the wave's buffers are released

    try:
        # Waves of len(pool) pairs, software-pipelined: the analysis of wave w + 1 is enqueued (same streams, behind
        # wave w's analysis) BEFORE the host turns to wave w's alignment, whose two read-backs per pair (trim lengths,
        # row count) would otherwise leave the GPU idle.
        prev = None
        for w0 in range(0, len(pairs), len(pool)):
            wave = []
            for k in range(len(pairs[w0:w0 + len(pool)])):
                src, tgt = uploads.get()        # (the pair's arrays, on the device already)
                ready = None
                if silence_for is not None:
                    sil = silence_for(w0 + k)
                elif rng is not None:
                    with torch.cuda.stream(rng.stream):
                        sil = rng.abs_normal_blocks(scale, [torch.empty((PAD_LEN, K), dtype=torch.float64, device=dev)
                                                            for _ in range(4)])
                    ready = rng.record_event()
                else:
                    sil = ahead.get()
                wave.append(TrainPair(device_index, fs, src, tgt, order=order, radius=radius,
                                      frame_period=frame_period, stream=pool.streams[k], ctx=pool.contexts[k],
                                      silence=sil, silence_ready=ready))
            for p in wave:
                p.analyse()
            if prev is not None:
                finish(prev)
            prev = wave
        if prev is not None:
            finish(prev)
        torch.cuda.synchronize(dev)
    finally:
        if ahead is not None:
            ahead.stop()
        if uploads is not None:
            uploads.stop()
    if not blocks:
        return torch.empty((0, 6 * order), dtype=torch.float64, device=dev), 0
    X = torch.cat(blocks).contiguous()
    torch.cuda.current_stream(dev).synchronize()      # the fit reads X on its own stream
    return X, frames


def fit_converter(X, components=64, seed=None, max_iter=100, device_index=0, verbose=0):
    """GMMFeatureConverter._train on the device-resident matrix (this rank's shard)."""
    from .converter.gmm_fit import GaussianMixtureHIP
    return GaussianMixtureHIP(n_components=components, max_iter=max_iter, random_state=seed, verbose=verbose,
                              device_index=device_index).fit(X)


def _stream_batch(make_pipeline, utterances, pool, shapes_per_stream, keep):
    """Utterance i on stream i % len(pool); a stream keeps the pipelines (buffers) of its `shapes_per_stream` most
    recently used utterance shapes.  A shape seen for the first time just runs kernel by kernel; from its SECOND
    appearance on the stream its pass is a captured HIP graph -- a corpus of files of all different lengths pays
    neither the extra passes of a capture nor the memory of a graph per file.  All pipelines of a stream share the
    stream's context and its scratch arena, which moves when a longer utterance needs more room: a graph captured
    before such a move is discarded and captured again (`_Graphed.graph_valid`) instead of being replayed against
    freed memory.  keep(i, pipeline): called with the pipeline's stream current right after utterance i was enqueued.
    The streams run independently of each other and the host waits once, at the end."""
    from collections import OrderedDict
    cache = [OrderedDict() for _ in range(len(pool))]
    for i, u in enumerate(utterances):
        k = i % len(pool)
        shape = (len(u[0]), len(u[1]))
        p = cache[k].get(shape)
        if p is None:
            while len(cache[k]) >= max(1, shapes_per_stream):
                _, old = cache[k].popitem(last=False)
                old.sync()                  # its buffers go back to the allocator: nothing of it may still be queued
            p = make_pipeline(u, pool.streams[k], pool.contexts[k])
            cache[k][shape] = p
            p.run()
        else:
            cache[k].move_to_end(shape)
            p.load(u)
            if not p.graph_valid():
                p.capture()                 # a plain pass (sizes the arena for this shape), then the capture
            p.replay()
        with torch.cuda.stream(p.stream):
            keep(i, p)
    for s_ in pool.streams:
        s_.synchronize()


def convert_batch(utterances, fs, gmm, device_index=0, order=24, frame_period=5.0, streams=16, pool=None,
                  shapes_per_stream=4, driver=None, lockstep=None, pcm=False, diff=False):
    """Convert this rank's utterances with the fitted mixture: list of waveforms (device tensors).
    Lockstep driver only: an utterance may be a bare waveform (its f0 is then extracted on the device), and pcm=True
    returns (waveforms, int16 tensors of the post-processed samples) -- wav in, 16-bit PCM out without the host;
    diff=True appends the differential outputs (the inputs through the MLSA filter of the differential conversion,
    convert_voice.py's .diff.wav): (waveforms, pcm or None, diff waveforms, diff pcm or None).
    driver='lockstep' (default): waves of 16 utterances through the batched entries on two streams (`ConvertWave`);
    'streams': round 3's utterance-per-stream driver, see `_stream_batch` for its scheduling."""
    dev = torch.device('cuda', device_index)
    dg = DeviceGMM(gmm.weights_, gmm.means_, gmm.covariances_, dev)
    out = [None] * len(utterances)
    driver = driver or ('streams' if pool is not None else 'lockstep')     # (a caller's pool asks for the stream driver)
    if driver == 'lockstep':
        pcms, dwav, dpcm = ([None] * len(utterances) for _ in range(3))

        def keep_view(i, w, p=None, wd=None, pd=None):
            out[i], pcms[i], dwav[i], dpcm[i] = w, p, wd, pd      # (views of their wave's blocks, which live as long as the views)
        _lockstep_batch(utterances, fs, device_index, dg, order, frame_period, lockstep, keep_view, pcm=pcm, diff=diff)
        if diff:
            return out, (pcms if pcm else None), dwav, (dpcm if pcm else None)
        return (out, pcms) if pcm else out
    if pcm or diff or (len(utterances) and not isinstance(utterances[0], (tuple, list))):
        raise ValueError('convert_batch: wav-in utterances and pcm=True need the lockstep driver')
    if pool is None:
        pool = StreamPool(device_index, streams)

    def keep(i, p):
        out[i] = p.wave.clone()
    _stream_batch(lambda u, st, ctx: ConvertPipeline(device_index, fs, u, dg, order=order, frame_period=frame_period,
                                                     stream=st, ctx=ctx), utterances, pool, shapes_per_stream, keep)
    return out


def resynthesize_batch(utterances, fs, device_index=0, frame_period=5.0, streams=16, pool=None, shapes_per_stream=4,
                       out=None, driver=None, lockstep=None):
    """BASELINE config 4 on one rank: analyse + resynthesise every utterance ((x, f0, t) triples: numpy arrays or
    device tensors), more utterances than the driver has in flight.  Returns the list of waveforms (device tensors;
    written into `out[i]` instead when a list of preallocated tensors is given) and the number of frames analysed.
    driver='lockstep' (default): waves of 16 through the batched entries; 'streams': a fixed pool of streams."""
    res = [None] * len(utterances)
    frames = 0
    for u in utterances:
        frames += len(u[1])
    driver = driver or ('streams' if pool is not None else 'lockstep')
    if driver == 'lockstep':
        def keep_view(i, w):
            if out is not None:
                out[i].copy_(w)
                res[i] = out[i]
            else:
                res[i] = w
        _lockstep_batch(utterances, fs, device_index, None, 24, frame_period, lockstep, keep_view)
        return res, frames
    from .pipeline import UtterancePipeline
    if pool is None:
        pool = StreamPool(device_index, streams)

    def keep(i, p):
        if out is not None:
            out[i].copy_(p.wave)
            res[i] = out[i]
        else:
            res[i] = p.wave.clone()
    _stream_batch(lambda u, st, ctx: UtterancePipeline(device_index, fs, u, frame_period=frame_period, stream=st,
                                                       ctx=ctx), utterances, pool, shapes_per_stream, keep)
    return res, frames
