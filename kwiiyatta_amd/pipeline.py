"""HBM-resident batch pipeline for the per-utterance conversion hot path.

The Python API of the package (kwiiyatta_amd.analyze_wav / align / convert /
synthesize) takes and returns numpy arrays like the reference, so every stage
crosses PCIe.  For batch conversion the whole source/target pair stays on the
device instead: one `PairPipeline` owns the buffers of one pair and one HIP
stream (a `kwy_ctx`), and `run()` only ENQUEUES kernels -- no host
synchronisation, no allocation -- so several pipelines overlap on one GPU
(utterance-per-stream) and ranks shard pairs with no collective
(utterance-per-GPU).

Stage order (the flow of `kwiieiya --carrier`
followed by `kwiiyatta`'s conversion):

  analyse source and target      CheapTrick + D4C (f0 tracks given)
  pad 100 silent frames          kwiiyatta.pad_silence
  sp2mc                          mel-cepstra of both padded features
  DTW features + FastDTW         make_feature(vuv='f0'), radius 32
  project path, gather rows      source frames on the target's time axis
  convert                        delta + GMM posterior + MLPG (c0 kept)
  mc2sp + WORLD synthesis        with the target's f0

torch is used for device memory, streams and strided copies only.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import lib, c_vp

EPS = 2.220446049250313e-16
SAFE_GUARD_MINIMUM = 1e-12
PAD_LEN = 100
POWER_WEIGHT, POWER_THRESHOLD, VUV_WEIGHT = 9.4, 1.636, 9.0
PIECE_CEILING = float(np.power(10, -1 / 10))          # normalize_data's default peak_lv = -1 (kwiiyatta/wavfile.py:8-12)


def _p(t):
    return c_vp(t.data_ptr())


class DeviceGMM:
    """Joint GMM parameters resident in HBM (weights (M,), means (M, 2D), covs (M, 2D, 2D))."""

    def __init__(self, weights, means, covs, device):
        self.M = len(weights)
        self.D2 = means.shape[1]
        self.weights = torch.from_numpy(np.ascontiguousarray(weights, dtype=np.float64)).to(device)
        self.means = torch.from_numpy(np.ascontiguousarray(means, dtype=np.float64)).to(device)
        self.covs = torch.from_numpy(np.ascontiguousarray(covs, dtype=np.float64)).to(device)
        self.device = torch.device(device)
        self._model = {}

    def model(self, diff=False):
        """The per-mixture matrices MLPG needs (Cholesky factor of Sxx, A = Syx Sxx^-1, conditional
        variances ...), computed on the GPU once per GMM instead of once per converted utterance."""
        diff = bool(diff)
        if diff not in self._model:
            d = self.D2 // 6
            buf = torch.empty(lib.kwy_gmm_model_doubles(d, self.M), dtype=torch.float64, device=self.device)
            ctx = _lib.Context(self.device.index or 0, stream=torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(ctx, lib.kwy_gmm_prepare_dev(ctx.handle, _p(self.weights), _p(self.means), _p(self.covs),
                                                    d, self.M, int(diff), _p(buf)))
            self._model[diff] = buf
        return self._model[diff]


class _Side:
    """Buffers of one analysed utterance, stored with PAD_LEN silent frames on both ends."""

    def __init__(self, x, f0, t, fs, K, order, dev):
        self.N, self.T = len(x), len(f0)
        self.Tp = self.T + 2 * PAD_LEN
        f64 = dict(dtype=torch.float64, device=dev)
        self.x = torch.from_numpy(x).to(dev)
        self.f0 = torch.from_numpy(f0).to(dev)
        self.t = torch.from_numpy(t).to(dev)
        self.f0_pad = torch.zeros(self.Tp, **f64)
        self.f0_pad[PAD_LEN:PAD_LEN + self.T] = self.f0
        self.sp_pad = torch.zeros((self.Tp, K), **f64)
        self.ap_pad = torch.full((self.Tp, K), 1 - SAFE_GUARD_MINIMUM, **f64)
        self.mc_pad = torch.empty((self.Tp, order + 1), **f64)
        self.feat = torch.empty((self.Tp, order + 2), **f64)
        # views of the un-padded middle part (the kernels write straight into them)
        self.sp = self.sp_pad[PAD_LEN:PAD_LEN + self.T]
        self.ap = self.ap_pad[PAD_LEN:PAD_LEN + self.T]

    def silence_rows(self):
        return self.sp_pad[:PAD_LEN], self.sp_pad[PAD_LEN + self.T:]


class _Graphed:
    """One pass of a pipeline as a HIP graph: `capture()` records the launches of `run()` once
    (after a plain pass has filled the library's tables and sized its arena), `replay()` enqueues
    them with a single call.  The pass has no host synchronisation and no data-dependent launch
    sizes -- every size that depends on the data (pulse counts, path length) lives on the device."""
    graph = None

    def capture(self, profile=False):
        """profile=True: the library's HIP events around its tracked kernels are captured with them; every
        replay records them again, and `ctx.profile_read` (once, when the graph is no longer used) returns the
        durations of the last replay."""
        self.run()                  # a plain pass first: tables, arena
        self.sync()
        g = torch.cuda.CUDAGraph()
        if profile:
            self.profile(True)
        with torch.cuda.graph(g, stream=self.stream):
            self.run()
        if profile:
            self.profile(False)
        self.graph = g
        self._arena_gen = [c.arena_generation() for c in self.contexts()]
        self.sync()

    def graph_valid(self):
        """False once a context's scratch arena has been relocated since the capture (another, larger call on the same
        context): the graph's kernel nodes then hold addresses of freed memory and it must be captured again."""
        return self.graph is not None and self._arena_gen == [c.arena_generation() for c in self.contexts()]

    def replay(self):
        if not self.graph_valid():
            raise RuntimeError('the captured graph is stale: the context\'s scratch arena moved after the capture '
                               '(capture() again, or give the pipeline a context of its own)')
        with torch.cuda.stream(self.stream):
            self.graph.replay()

    def contexts(self):
        """the library contexts this pipeline launches on (one per stream)"""
        return [c for c in (self.ctx, getattr(self, 'side_ctx', None)) if c is not None]

    def profile(self, enable=True):
        for c in self.contexts():
            c.profile(enable)

    def profile_read(self, kernel):
        """(summed milliseconds, launches) of `kernel` over the pipeline's streams since the last read"""
        ms, n = 0.0, 0
        for c in self.contexts():
            a, b = c.profile_read(kernel)
            ms += a
            n += b
        return ms, n


def draw_silence(fs, K, frame_len=PAD_LEN):
    """WorldSynthesizer._silence_spectrum_envelope (kwiiyatta/vocoder/world.py:158-161): |N(0, EPS / fs)| from
    numpy's GLOBAL legacy generator, which the reference's tests seed"""
    return np.abs(np.random.normal(0, EPS / fs, (frame_len, K)))


class PairPipeline(_Graphed):
    def __init__(self, device_index, fs, source, target, gmm, order=24, radius=32, frame_period=5.0,
                 stream=None, prepare_gmm_per_run=False, silence=None, keep_aligned_spectrum=False,
                 side_stream=False):
        """source / target: (x, f0, timeaxis) numpy triples; gmm: DeviceGMM over 2*3*order dims.
        prepare_gmm_per_run: redo the GMM-only part of MLPG (nnmnkwii's MLPG.__init__) in every run(),
        as the reference does per convert() call, instead of once per converter.
        silence: the four (100, K) silent spectra of pad_silence (source head, source tail, target head, target
        tail); default: drawn here from numpy's global generator in that order, as `align` would -- uploaded once,
        so the pipeline and the Python API path see the same pads under `np.random.seed`.
        keep_aligned_spectrum: also gather the aligned source spectrum (`align` returns it; the conversion flow
        replaces it by the converted one and never reads it).
        side_stream: a second stream takes the target's CheapTrick -> sp2mc -> DTW features and then D4C of both
        utterances (which needs only waveform, f0 and frame times), while the first does the source's envelope and
        features and starts FastDTW as soon as the target's features are there; joined before the aligned
        aperiodicity is gathered.  The alignment is a chain of single-workgroup kernels, so D4C overlaps with it
        almost completely; the synthesis' pulse placement (f0 only) runs there too.  A latency option (one pair alone:
        3.3 -> 2.4 ms); with many pairs in flight the extra
        streams cost throughput (32 pairs: 1.49 M -> 1.30 M frames/s), hence off by default."""
        self.prepare_gmm_per_run = bool(prepare_gmm_per_run)
        self.keep_aligned_spectrum = bool(keep_aligned_spectrum)
        self.dev = torch.device('cuda', device_index)
        self.fs, self.order, self.radius, self.frame_period = int(fs), int(order), int(radius), float(frame_period)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.dev)
        self.ctx = _lib.Context(device_index, stream=self.stream.cuda_stream)
        self.side = torch.cuda.Stream(device=self.dev) if side_stream else None
        self.side_ctx = _lib.Context(device_index, stream=self.side.cuda_stream) if side_stream else None
        self.fft = lib.kwy_cheaptrick_fft_size(self.fs, 71.0)
        self.K = self.fft // 2 + 1
        from .backend import sptk
        self.alpha = sptk.mcepalpha(self.fs)
        self.gmm = gmm
        assert gmm.D2 == 6 * order
        self.gmm_model = gmm.model(diff=False)
        f64 = dict(dtype=torch.float64, device=self.dev)
        with torch.cuda.stream(self.stream):
            self.src = _Side(*source, self.fs, self.K, order, self.dev)
            self.tgt = _Side(*target, self.fs, self.K, order, self.dev)
            Tt = self.tgt.T
            cap = self.src.Tp + self.tgt.Tp + 2
            self.path = torch.zeros((cap, 2), dtype=torch.int32, device=self.dev)
            self.path_len = torch.zeros(1, dtype=torch.int64, device=self.dev)
            self.dist = torch.zeros(1, **f64)
            self.idx = torch.zeros(self.tgt.Tp, dtype=torch.int32, device=self.dev)
            self.n_idx = torch.zeros(1, dtype=torch.int64, device=self.dev)
            self.sp_al = torch.empty((Tt, self.K), **f64) if self.keep_aligned_spectrum else None
            self.ap_al = torch.empty((Tt, self.K), **f64)
            self.mc_al = torch.empty((Tt, order + 1), **f64)
            self.mc_x = torch.empty((Tt, order), **f64)
            self.mc_y = torch.empty((Tt, order), **f64)
            self.mc_conv = torch.empty((Tt, order + 1), **f64)
            self.sp_conv = torch.empty((Tt, self.K), **f64)
            self.ylen = lib.kwy_synth_length(Tt, self.frame_period, self.fs)
            self.wave = torch.empty(self.ylen, **f64)
            # pulse placement of the synthesis (f0 only): computed on the side stream, ahead of the features
            self.plan = torch.empty(lib.kwy_synth_plan_bytes(self.ylen), dtype=torch.uint8, device=self.dev) \
                if self.side is not None else None
            # kwiiyatta.pad_silence: |N(0, EPS/fs)| spectra on the padding frames, host-drawn, uploaded once
            if silence is None:
                silence = [draw_silence(self.fs, self.K) for _ in range(4)]
            rows = self.src.silence_rows() + self.tgt.silence_rows()
            for dst, sil in zip(rows, silence):
                dst.copy_(torch.from_numpy(np.ascontiguousarray(sil)))
        self.stream.synchronize()
        self.frames = self.src.T   # the metric counts source frames

    def _chk(self, rc):
        _lib.check(self.ctx, rc)

    def run(self):
        """Enqueue one pass of the hot path on this pipeline's stream (asynchronous)."""
        h, fs, fft, K, order = self.ctx.handle, self.fs, self.fft, self.K, self.order
        def envelope(ctx, s):          # CheapTrick -> mel-cepstrum -> DTW feature rows of one utterance
            hh = ctx.handle
            _lib.check(ctx, lib.kwy_cheaptrick_dev(hh, _p(s.x), s.N, fs, _p(s.t), _p(s.f0), s.T, -0.15, 71.0, fft,
                                                   float(fs), _p(s.sp)))
            return hh

        def features(ctx, s):
            hh = ctx.handle
            _lib.check(ctx, lib.kwy_sp2mc_dev(hh, _p(s.sp_pad), s.Tp, K, order, self.alpha, _p(s.mc_pad)))
            _lib.check(ctx, lib.kwy_align_features_dev(hh, _p(s.mc_pad), s.Tp, order + 1, _p(s.f0_pad), POWER_WEIGHT,
                                                       POWER_THRESHOLD, VUV_WEIGHT, _p(s.feat)))

        def d4c(ctx, s):
            _lib.check(ctx, lib.kwy_d4c_dev(ctx.handle, _p(s.x), s.N, fs, _p(s.t), _p(s.f0), s.T, 0.85, fft, _p(s.ap)))

        with torch.cuda.stream(self.stream):
            if self.side is not None:
                # fork (inside a capture: a second branch of the graph).  Side stream: the target's envelope and DTW
                # features, then both aperiodicities; main stream: the source's envelope and features, then the
                # alignment as soon as the target's features are there.
                self.side.wait_stream(self.stream)
                with torch.cuda.stream(self.side):
                    envelope(self.side_ctx, self.tgt)
                    features(self.side_ctx, self.tgt)
                    target_ready = torch.cuda.Event()
                    target_ready.record(self.side)
                    d4c(self.side_ctx, self.src)
                    d4c(self.side_ctx, self.tgt)
                    _lib.check(self.side_ctx, lib.kwy_synth_plan_dev(self.side_ctx.handle, _p(self.tgt.f0), self.tgt.T,
                                                                     fft, self.frame_period, fs, self.ylen,
                                                                     _p(self.plan)))
                envelope(self.ctx, self.src)
                features(self.ctx, self.src)
                self.stream.wait_event(target_ready)
            else:
                # both utterances of the pair in one grid per kernel (kwy_*_batch_dev): 4 202 frames fill the chip
                # better than 2 001 + 2 201 one after the other, and half the launches
                both = (self.src, self.tgt)
                _lib.check(self.ctx, lib.kwy_cheaptrick_batch_dev(
                    h, _lib.utterance_array([(s.x, s.t, s.f0, s.sp) for s in both]), 2, fs, -0.15, 71.0, fft, float(fs)))
                _lib.check(self.ctx, lib.kwy_d4c_batch_dev(
                    h, _lib.utterance_array([(s.x, s.t, s.f0, s.ap) for s in both]), 2, fs, 0.85, fft))
                for s in both:
                    features(self.ctx, s)
            self._chk(lib.kwy_fastdtw_dev(h, _p(self.src.feat), self.src.Tp, _p(self.tgt.feat), self.tgt.Tp,
                                          order + 2, self.radius, _p(self.dist), _p(self.path),
                                          _p(self.path_len)))
            self._chk(lib.kwy_align_project_dev(h, _p(self.path), _p(self.path_len), PAD_LEN, _p(self.idx),
                                                self.tgt.Tp, _p(self.n_idx)))
            Tt = self.tgt.T
            if self.side is not None:
                self.stream.wait_stream(self.side)              # join: the aperiodicities are complete
            for src_arr, dst, w in ((self.src.sp_pad, self.sp_al, K), (self.src.ap_pad, self.ap_al, K),
                                    (self.src.mc_pad, self.mc_al, order + 1)):
                if dst is not None:
                    self._chk(lib.kwy_gather_rows_dev(h, _p(src_arr), self.src.Tp, w, _p(self.idx), Tt, _p(dst)))
            g = self.gmm
            if self.prepare_gmm_per_run:
                self.mc_x.copy_(self.mc_al[:, 1:])
                self._chk(lib.kwy_gmm_mlpg_dev(h, _p(self.mc_x), Tt, order, g.M, _p(g.weights), _p(g.means),
                                               _p(g.covs), 0, _p(self.mc_y)))
                self.mc_conv[:, 0].copy_(self.mc_al[:, 0])
                self.mc_conv[:, 1:].copy_(self.mc_y)
            else:       # c0 kept, c1.. converted, in one call on the strided rows
                self._chk(lib.kwy_convert_mcep_dev(h, _p(self.mc_al), Tt, order, g.M, _p(self.gmm_model),
                                                   _p(self.mc_conv)))
            self._chk(lib.kwy_mc2sp_dev(h, _p(self.mc_conv), Tt, order, self.alpha, fft, _p(self.sp_conv)))
            if self.side is not None:
                self._chk(lib.kwy_synth_render_dev(h, _p(self.plan), Tt, _p(self.sp_conv), _p(self.ap_al), fft,
                                                   self.frame_period, fs, float(fs), self.ylen, _p(self.wave)))
            else:
                self._chk(lib.kwy_synthesize_dev(h, _p(self.tgt.f0), Tt, _p(self.sp_conv), _p(self.ap_al), fft,
                                                 self.frame_period, fs, float(fs), self.ylen, _p(self.wave)))

    def sync(self):
        self.ctx.sync()


class _Wave:
    """The pairs of one wave of a PairBatchPipeline: two streams with a library context each, and the wave's buffers
    as CONTIGUOUS blocks -- the padded features of all its utterances form one (rows, K) matrix, so the row-wise
    stages (sp2mc, mc2sp) are one launch over the wave, and the ragged stages take per-utterance views of it."""

    def __init__(self, owner, pairs, main_stream=None, serial=False):
        dev, K, order, fs = owner.dev, owner.K, owner.order, owner.fs
        f64 = dict(dtype=torch.float64, device=dev)
        self.n = len(pairs)
        # the stream of the serial chain (alignment, conversion, rendering: narrow kernels between chip-wide ones) may
        # be given priority over the aperiodicity stream (owner.chain_priority, bench.py --chain-priority)
        self.stream = main_stream if main_stream is not None else \
            torch.cuda.Stream(device=dev, priority=-1 if owner.chain_priority else 0)
        self.ctx = _lib.Context(dev.index, stream=self.stream.cuda_stream)
        if serial:          # one stream for everything: kernels one after the other (per-kernel timing, rocprofv3 runs)
            self.side, self.side_ctx = self.stream, self.ctx
        else:
            self.side = torch.cuda.Stream(device=dev)
            self.side_ctx = _lib.Context(dev.index, stream=self.side.cuda_stream)
        sides = [s for pair in pairs for s in pair]               # source 0, target 0, source 1, ...
        wav_in = owner.wav_in
        sides = [(s,) if wav_in and not isinstance(s, (tuple, list)) else s for s in sides]
        self.N = [len(s[0]) for s in sides]
        # wav in: the frame grid is DIO's (kwy_dio_frames), the f0 track is extracted inside the step
        self.T = [int(lib.kwy_dio_frames(fs, len(s[0]), owner.frame_period)) for s in sides] if wav_in else \
            [len(s[1]) for s in sides]
        self.Tp = [t + 2 * PAD_LEN for t in self.T]
        cat = lambda k: torch.from_numpy(np.concatenate([np.ascontiguousarray(s[k], dtype=np.float64)  # noqa: E731
                                                         for s in sides])).to(dev)
        with torch.cuda.stream(self.stream):
            self.x_all = cat(0)
            if wav_in:
                self.t_all = torch.empty(int(sum(self.T)), **f64)
                self.f0_all = torch.empty(int(sum(self.T)), **f64)        # DIO's track, before the refinement
            else:
                self.f0_all, self.t_all = cat(1), cat(2)
            rows = int(sum(self.Tp))
            fused = owner.fused_mcep
            if not fused:
                self.sp_pad = torch.zeros((rows, K), **f64)
            self.ap_pad = torch.full((rows, K), 1 - SAFE_GUARD_MINIMUM, **f64)
            self.f0_pad = torch.zeros(rows, **f64)
            self.mc_pad = torch.empty((rows, order + 1), **f64)
            self.feat = torch.empty((rows, order + 2), **f64)
            xo = np.concatenate(([0], np.cumsum(self.N)))
            to = np.concatenate(([0], np.cumsum(self.T)))
            po = np.concatenate(([0], np.cumsum(self.Tp)))
            cut = lambda a, o, i, lo=0, hi=0: a[int(o[i]) + lo:int(o[i + 1]) - hi]  # noqa: E731
            ns = len(sides)
            self.x = [cut(self.x_all, xo, i) for i in range(ns)]
            self.f0 = [cut(self.f0_all, to, i) for i in range(ns)]
            self.t = [cut(self.t_all, to, i) for i in range(ns)]
            self.ap = [cut(self.ap_pad, po, i, PAD_LEN, PAD_LEN) for i in range(ns)]
            if not fused:
                self.sp = [cut(self.sp_pad, po, i, PAD_LEN, PAD_LEN) for i in range(ns)]      # the un-padded middle parts
                self.sp_p = [cut(self.sp_pad, po, i) for i in range(ns)]
            self.ap_p = [cut(self.ap_pad, po, i) for i in range(ns)]
            self.mc_p = [cut(self.mc_pad, po, i) for i in range(ns)]
            self.f0_p = [cut(self.f0_pad, po, i) for i in range(ns)]
            self.feat_p = [cut(self.feat, po, i) for i in range(ns)]
            if wav_in:
                # StoneMask writes the refined track straight into the padded f0 rows; every consumer reads it there
                self.f0_dio = self.f0
                self.f0 = [self.f0_p[i][PAD_LEN:PAD_LEN + self.T[i]] for i in range(ns)]
                self.f0_status = torch.zeros(ns, dtype=torch.int32, device=dev)
                self.j_dio = _lib.job_array(_lib.F0Job, [(self.x[i], self.N[i], self.t[i], self.f0_dio[i],
                                                          self.f0_status[i:i + 1]) for i in range(ns)])
                self.j_sm = _lib.utterance_array([(self.x[i], self.t[i], self.f0_dio[i], self.f0[i]) for i in range(ns)])
            else:
                for i in range(ns):
                    self.f0_p[i][PAD_LEN:PAD_LEN + self.T[i]] = self.f0[i]
            # pad rows in the reference's order of draws: source head, source tail, target head, target tail
            if not fused:
                self.pad_rows = [blk for i in range(ns) for blk in (self.sp_p[i][:PAD_LEN], self.sp_p[i][PAD_LEN + self.T[i]:])]
            else:
                # No envelope rows at all: CheapTrick hands over mel-cepstra (kwy_cheaptrick_mcep_batch_dev) straight into
                # the padded rows' middle parts; the pad SPECTRA live in one block of their own, their mel-cepstra
                # (sp2mc over 2 x 100 rows per side) are copied to the rows around
                self.pads_sp = torch.zeros((2 * ns * PAD_LEN, K), **f64)
                self.pads_mc = torch.empty((2 * ns * PAD_LEN, order + 1), **f64)
                self.pad_rows = [self.pads_sp[b * PAD_LEN:(b + 1) * PAD_LEN] for b in range(2 * ns)]
                self.pad_idx = torch.arange(PAD_LEN, dtype=torch.int32, device=dev)
                pm = lambda b: self.pads_mc[b * PAD_LEN:(b + 1) * PAD_LEN]  # noqa: E731
                self.j_padmc = _lib.job_array(_lib.GatherJob, [row for i in range(ns) for row in (
                    (pm(2 * i), PAD_LEN, self.pad_idx, PAD_LEN, self.mc_p[i][:PAD_LEN]),
                    (pm(2 * i + 1), PAD_LEN, self.pad_idx, PAD_LEN, self.mc_p[i][PAD_LEN + self.T[i]:]))])
            # per pair, on the target's time axis
            Tt = [self.T[2 * k + 1] for k in range(self.n)]
            self.Tt = Tt
            ao = np.concatenate(([0], np.cumsum(Tt)))
            arows = int(ao[-1])
            self.ap_al = torch.empty((arows, K), **f64)
            self.mc_al = torch.empty((arows, order + 1), **f64)
            self.mc_conv = torch.empty((arows, order + 1), **f64)
            self.sp_conv = torch.empty((arows, K), **f64)
            self.ylen = [lib.kwy_synth_length(t, owner.frame_period, fs) for t in Tt]
            yo = np.concatenate(([0], np.cumsum(self.ylen)))
            self.wave_all = torch.empty(int(yo[-1]), **f64)
            self.wave = [cut(self.wave_all, yo, k) for k in range(self.n)]
            if owner.pcm:         # the post-step and the 16-bit samples on the device (kwy_finish_pcm16_batch_dev)
                self.pcm_all = torch.zeros(int(yo[-1]), dtype=torch.int16, device=dev)
                self.pcm = [cut(self.pcm_all, yo, k) for k in range(self.n)]
                self.j_fin = _lib.job_array(_lib.FinishJob, [(self.wave[k], self.ylen[k], Tt[k], self.pcm[k])
                                                             for k in range(self.n)])
            cap = [self.Tp[2 * k] + self.Tp[2 * k + 1] + 2 for k in range(self.n)]
            self.path = [torch.zeros((c, 2), dtype=torch.int32, device=dev) for c in cap]
            self.path_len = torch.zeros(self.n, dtype=torch.int64, device=dev)
            self.dist = torch.zeros(self.n, **f64)
            self.idx = [torch.zeros(self.Tp[2 * k + 1], dtype=torch.int32, device=dev) for k in range(self.n)]
            self.n_idx = torch.zeros(self.n, dtype=torch.int64, device=dev)
            self.plan = [torch.empty(lib.kwy_synth_plan_bytes(y), dtype=torch.uint8, device=dev) for y in self.ylen]
            one = lambda a, k: a[k:k + 1]  # noqa: E731
            J = _lib.job_array
            both = range(ns)
            self.j_env = _lib.utterance_array([(self.x[i], self.t[i], self.f0[i],
                                                self.mc_p[i][PAD_LEN:PAD_LEN + self.T[i]] if fused else self.sp[i]) for i in both])
            self.j_ap = _lib.utterance_array([(self.x[i], self.t[i], self.f0[i], self.ap[i]) for i in both])
            self.j_feat = J(_lib.AlignJob, [(self.mc_p[i], self.f0_p[i], self.Tp[i], self.feat_p[i]) for i in both])
            self.j_dtw = J(_lib.DtwJob, [(self.feat_p[2 * k], self.Tp[2 * k], self.feat_p[2 * k + 1], self.Tp[2 * k + 1],
                                          one(self.dist, k), self.path[k], one(self.path_len, k)) for k in range(self.n)])
            self.j_proj = J(_lib.ProjectJob, [(self.path[k], one(self.path_len, k), self.idx[k], self.Tp[2 * k + 1],
                                               one(self.n_idx, k)) for k in range(self.n)])
            self.j_gap = J(_lib.GatherJob, [(self.ap_p[2 * k], self.Tp[2 * k], self.idx[k], Tt[k], cut(self.ap_al, ao, k))
                                            for k in range(self.n)])
            self.j_gmc = J(_lib.GatherJob, [(self.mc_p[2 * k], self.Tp[2 * k], self.idx[k], Tt[k], cut(self.mc_al, ao, k))
                                            for k in range(self.n)])
            self.j_conv = J(_lib.ConvertJob, [(cut(self.mc_al, ao, k), Tt[k], cut(self.mc_conv, ao, k)) for k in range(self.n)])
            self.j_plan = J(_lib.SynthPlanJob, [(self.f0[2 * k + 1], Tt[k], self.ylen[k], self.plan[k]) for k in range(self.n)])
            self.j_render = _lib.synth_job_array([(self.plan[k], cut(self.sp_conv, ao, k), cut(self.ap_al, ao, k), self.wave[k])
                                                  for k in range(self.n)])
        self.rows, self.arows = rows, arows
        self.frames = sum(self.T[0::2])


class PairBatchPipeline(_Graphed):
    """The hot path of a BATCH of pairs in lockstep: every stage is one launch (or one pass of launches) over all pairs
    of a wave -- the batched entries of include/kwy.h -- instead of one stream per pair.

    The serial stages of a pair (FastDTW's recurrence, the synthesis' phase chain, the trajectory solve) are one
    workgroup each; with N pairs in one grid they occupy N compute units side by side, and the chip-wide stages run over
    N times the frames.  A wave (<= 16 pairs: the batch limit of one launch) uses two streams:

        main   CheapTrick of both sides -> [pads] -> sp2mc -> DTW features -> FastDTW -> projection
               -> [aperiodicity] -> gathers -> conversion -> mc2sp -> [plan] -> rendering
        side   D4C of both sides (waveform, f0 and frame times only) -> pulse placement (target f0 only)

    and `waves` waves run side by side (default 2: four streams, the number of hardware queues the HIP runtime creates
    by default -- no GPU_MAX_HW_QUEUES needed).  One wave's serial kernels overlap with the other waves' chip-wide
    ones.  The whole step is one HIP graph (`capture()` / `replay()`).

    rng: a backend.nprandom.DeviceRandomState.  Every run then draws FRESH pad spectra for all pairs from numpy's
    legacy stream -- source head, source tail, target head, target tail, pair after pair: the draws `align` makes
    (kwiiyatta/vocoder/feature.py:19-41, world.py:158-161) -- in one call at the head of the step.  Without it the
    pads are `silence` (4 blocks per pair) or drawn here once from numpy's global generator, as PairPipeline does.

    rng_place: where the draw runs -- 'side' (default: at the head of the first wave's side stream, in front of its
    D4C; measured at 32 pairs: 29.7 ms per step against 29.4 without any draw), 'head' (on the main stream before the
    fork: 30.0), 'own' (the generator's own stream as a fifth branch of the graph: 31.8).
    serial: everything on ONE stream, kernel after kernel (for per-kernel timing).

    wav_in: the pairs are bare waveforms (or (x, ...) tuples whose f0 is ignored): DIO + StoneMask of both sides run at
    the head of every wave (kwy_dio_batch_dev, kwy_stonemask_batch_dev: Analyzer.extract_f0,
    kwiiyatta/vocoder/world.py:33-41), the refined track going straight into the padded f0 rows.  pcm: the post-step
    and the int16 samples of every waveform at the end of the wave (kwy_finish_pcm16_batch_dev:
    vocoder/abc/synthesizer.py:11-20 + wavfile.py:8-29), `pcm(k)`.

    fused_mcep (default): CheapTrick hands over mel-cepstra (kwy_cheaptrick_mcep_batch_dev: its liftered cepstrum
    through pysptk's frequency transform -- no envelope row is written, no sp2mc re-reads it); only the pad spectra go
    through sp2mc.  False: the two calls of the reference's stages, bit-equal to PairPipeline.

    Outputs per pair k: `wave(k)`; with fused_mcep=False equal to PairPipeline's bit for bit given the same pads, with
    it within the rounding of one exp / log round trip (same DTW path in the tests, waveforms within 1e-9)."""

    def __init__(self, device_index, fs, pairs, gmm, order=24, radius=32, frame_period=5.0, waves=2, rng=None,
                 silence=None, rng_place='side', serial=False, max_wave=16, wav_in=False, pcm=False, fused_mcep=True,
                 chain_priority=False):
        self.dev = torch.device('cuda', device_index)
        self.chain_priority = bool(chain_priority)
        self.wav_in, self.pcm, self.fused_mcep = bool(wav_in), bool(pcm), bool(fused_mcep)
        self.fs, self.order, self.radius, self.frame_period = int(fs), int(order), int(radius), float(frame_period)
        self.fft = lib.kwy_cheaptrick_fft_size(self.fs, 71.0)
        self.K = self.fft // 2 + 1
        from .backend import sptk
        self.alpha = sptk.mcepalpha(self.fs)
        self.gmm = gmm
        assert gmm.D2 == 6 * order
        self.gmm_model = gmm.model(diff=False)
        self.rng = rng
        self.rng_place = rng_place
        pairs = list(pairs)
        nw = max(1, min(int(waves), len(pairs)))
        while (len(pairs) + nw - 1) // nw > max_wave:  # (16: a wave is one launch of the batched entries)
            nw += 1
        per = (len(pairs) + nw - 1) // nw
        self.stream = torch.cuda.Stream(device=self.dev, priority=-1 if self.chain_priority else 0)
        self.waves = []
        for w in range(nw):
            chunk = pairs[w * per:(w + 1) * per]
            if chunk:
                self.waves.append(_Wave(self, chunk, main_stream=self.stream if (w == 0 or serial) else None, serial=serial))
        self.ctx = self.waves[0].ctx
        self.where = [(w, k) for w, wv in enumerate(self.waves) for k in range(wv.n)]
        self.pad_rows = [blk for wv in self.waves for blk in wv.pad_rows]
        if rng is None:
            if silence is None:
                silence = [draw_silence(self.fs, self.K) for _ in range(4 * len(pairs))]
            with torch.cuda.stream(self.stream):
                for dst, sil in zip(self.pad_rows, silence):
                    dst.copy_(torch.from_numpy(np.ascontiguousarray(sil)))
        for wv in self.waves:
            wv.stream.synchronize()
        self.frames = sum(wv.frames for wv in self.waves)
        self.n_pairs = len(pairs)

    def wave(self, k):
        w, i = self.where[k]
        return self.waves[w].wave[i]

    def pcm16(self, k):
        w, i = self.where[k]
        return self.waves[w].pcm[i]

    def f0_status(self):
        """wav_in: the DIO status words of all utterances (one read-back; non-zero = zero-crossing overflow)"""
        return torch.cat([wv.f0_status for wv in self.waves]).cpu().numpy()

    def path(self, k):
        """(path, path_len, dist) device tensors of pair k"""
        w, i = self.where[k]
        wv = self.waves[w]
        return wv.path[i], wv.path_len[i:i + 1], wv.dist[i:i + 1]

    def contexts(self):
        cs = []
        for wv in self.waves:
            for c in (wv.ctx, wv.side_ctx):
                if not any(c is o for o in cs):
                    cs.append(c)
        return cs + ([self.rng.ctx] if self.rng is not None and self.rng_place == 'own' else [])

    def run(self):
        fs, fft, K, order = self.fs, self.fft, self.K, self.order
        origin = self.stream
        pads = None
        with torch.cuda.stream(origin):
            if self.rng is not None and self.rng_place == 'own':
                # ONE draw for the step, on the generator's own stream: the pads are needed first by sp2mc, behind
                # CheapTrick, which the draw overlaps with
                self.rng.stream.wait_stream(origin)
                self.rng.abs_normal_blocks(EPS / fs, self.pad_rows)
                pads = self.rng.record_event()
            elif self.rng is not None and self.rng_place == 'head':
                self.rng.abs_normal_blocks(EPS / fs, self.pad_rows, ctx=self.ctx)
            if self.wav_in:
                # The f0 tracks of ALL waves first, on the origin stream, then the fork.  (With the extraction at the
                # head of every wave's own stream -- forks of the graph at two different depths -- hipStreamEndCapture
                # crashed on ROCm 7.2 as soon as there were two waves, whatever ran on the second one:
                # tools/capture_probe.py.  The kernels are chip-wide anyway.)
                c0 = self.waves[0].ctx
                for wv in self.waves:
                    _lib.check(c0, lib.kwy_dio_batch_dev(c0.handle, wv.j_dio, 2 * wv.n, fs, 71.0, 800.0, 2.0,
                                                         self.frame_period, 1, 0.1))
                    _lib.check(c0, lib.kwy_stonemask_batch_dev(c0.handle, wv.j_sm, 2 * wv.n, fs))
            for wv in self.waves:
                if wv.stream is not origin:
                    wv.stream.wait_stream(origin)
                wv.side.wait_stream(origin)
        for wv in self.waves:
            h, hs, n = wv.ctx.handle, wv.side_ctx.handle, wv.n
            with torch.cuda.stream(wv.side):
                if self.rng is not None and self.rng_place == 'side' and wv is self.waves[0]:
                    self.rng.abs_normal_blocks(EPS / fs, self.pad_rows, ctx=wv.side_ctx)
                    pads = torch.cuda.Event()
                    pads.record(wv.side)
                _lib.check(wv.side_ctx, lib.kwy_d4c_batch_dev(hs, wv.j_ap, 2 * n, fs, 0.85, fft))
                ap_done = torch.cuda.Event()
                ap_done.record(wv.side)
                _lib.check(wv.side_ctx, lib.kwy_synth_plan_batch_dev(hs, wv.j_plan, n, fft, self.frame_period, fs))
            with torch.cuda.stream(wv.stream):
                chk = lambda rc, c=wv.ctx: _lib.check(c, rc)  # noqa: E731
                if self.fused_mcep:
                    chk(lib.kwy_cheaptrick_mcep_batch_dev(h, wv.j_env, 2 * n, fs, -0.15, 71.0, fft, float(fs), order, self.alpha))
                    if pads is not None:
                        wv.stream.wait_event(pads)
                    chk(lib.kwy_sp2mc_dev(h, _p(wv.pads_sp), wv.pads_sp.shape[0], K, order, self.alpha, _p(wv.pads_mc)))
                    chk(lib.kwy_gather_rows_batch_dev(h, wv.j_padmc, 4 * n, order + 1))
                else:
                    chk(lib.kwy_cheaptrick_batch_dev(h, wv.j_env, 2 * n, fs, -0.15, 71.0, fft, float(fs)))
                    if pads is not None:
                        wv.stream.wait_event(pads)
                    chk(lib.kwy_sp2mc_dev(h, _p(wv.sp_pad), wv.rows, K, order, self.alpha, _p(wv.mc_pad)))
                chk(lib.kwy_align_features_batch_dev(h, wv.j_feat, 2 * n, order + 1, POWER_WEIGHT, POWER_THRESHOLD,
                                                     VUV_WEIGHT))
                chk(lib.kwy_fastdtw_batch_dev(h, wv.j_dtw, n, order + 2, self.radius))
                chk(lib.kwy_align_project_batch_dev(h, wv.j_proj, n, PAD_LEN))
                wv.stream.wait_event(ap_done)
                chk(lib.kwy_gather_rows_batch_dev(h, wv.j_gap, n, K))
                chk(lib.kwy_gather_rows_batch_dev(h, wv.j_gmc, n, order + 1))
                chk(lib.kwy_convert_mcep_batch_dev(h, wv.j_conv, n, order, self.gmm.M, _p(self.gmm_model)))
                chk(lib.kwy_mc2sp_dev(h, _p(wv.mc_conv), wv.arows, order, self.alpha, fft, _p(wv.sp_conv)))
                wv.stream.wait_stream(wv.side)                      # join: the plans are there
                chk(lib.kwy_synth_render_batch_dev(h, wv.j_render, n, fft, self.frame_period, fs, float(fs)))
                if self.pcm:
                    chk(lib.kwy_finish_pcm16_batch_dev(h, wv.j_fin, n, fs, 1, PIECE_CEILING, 1, PIECE_CEILING))
        with torch.cuda.stream(origin):
            for wv in self.waves:
                if wv.stream is not origin:
                    origin.wait_stream(wv.stream)
            if self.rng is not None and self.rng_place == 'own':
                origin.wait_stream(self.rng.stream)

    def sync(self):
        for wv in self.waves:
            wv.ctx.sync()
            wv.side_ctx.sync()


class UtterancePipeline(_Graphed):
    """analyse -> resynthesise of one utterance (BASELINE config 2; resynthesize_voice.py without a carrier,
    /root/reference/kwiiyatta/resynthesize_voice.py:46-79), HBM-resident."""

    def __init__(self, device_index, fs, utterance, frame_period=5.0, stream=None, ctx=None):
        self.dev = torch.device('cuda', device_index)
        self.fs, self.frame_period = int(fs), float(frame_period)
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.dev)
        self.ctx = ctx if ctx is not None else _lib.Context(device_index, stream=self.stream.cuda_stream)
        self.fft = lib.kwy_cheaptrick_fft_size(self.fs, 71.0)
        self.K = self.fft // 2 + 1
        x, f0, t = utterance
        self.N, self.T = len(x), len(f0)
        f64 = dict(dtype=torch.float64, device=self.dev)
        with torch.cuda.stream(self.stream):
            self.x, self.f0, self.t = (torch.empty(len(a), **f64) for a in (x, f0, t))
            self.sp = torch.empty((self.T, self.K), **f64)
            self.ap = torch.empty((self.T, self.K), **f64)
            self.ylen = lib.kwy_synth_length(self.T, self.frame_period, self.fs)
            self.wave = torch.empty(self.ylen, **f64)
        self.load(utterance)
        self.stream.synchronize()
        self.frames = self.T

    def load(self, utterance):
        """another utterance of the same shape into the existing buffers (numpy arrays or device tensors;
        asynchronous on the pipeline's stream)"""
        x, f0, t = utterance
        if len(x) != self.N or len(f0) != self.T:
            raise ValueError('UtterancePipeline.load: shape differs from the pipeline\'s')
        with torch.cuda.stream(self.stream):
            for dst, src in ((self.x, x), (self.f0, f0), (self.t, t)):
                if torch.is_tensor(src):
                    src.record_stream(self.stream)      # (the caller may drop it while the copy is still queued)
                dst.copy_(src if torch.is_tensor(src) else torch.from_numpy(np.ascontiguousarray(src)),
                          non_blocking=True)

    def run(self):
        h, fs, fft = self.ctx.handle, self.fs, self.fft
        chk = lambda rc: _lib.check(self.ctx, rc)  # noqa: E731
        with torch.cuda.stream(self.stream):
            chk(lib.kwy_cheaptrick_dev(h, _p(self.x), self.N, fs, _p(self.t), _p(self.f0), self.T, -0.15, 71.0, fft,
                                       float(fs), _p(self.sp)))
            chk(lib.kwy_d4c_dev(h, _p(self.x), self.N, fs, _p(self.t), _p(self.f0), self.T, 0.85, fft, _p(self.ap)))
            chk(lib.kwy_synthesize_dev(h, _p(self.f0), self.T, _p(self.sp), _p(self.ap), fft, self.frame_period, fs,
                                       float(fs), self.ylen, _p(self.wave)))

    def sync(self):
        self.ctx.sync()


class HostFeeder:
    """Waveforms from pinned host memory into a set of PairPipelines and the synthesised waveforms back, off the
    pipelines' own streams and in TWO transfers per step: all inputs of the step travel as one contiguous block on an
    upload stream, all outputs as one block on a download stream, through two staging slots -- the upload of step n+1
    and the download of step n-1 overlap with the kernels of step n.  The pipelines' input buffers keep their addresses
    (captured graphs stay valid): a pass starts with a device-to-device copy KERNEL out of the staging block (3.8 MB:
    microseconds of HBM time; copy-engine transfers in both directions spread over dozens of streams collapse to
    ~4 GB/s here, and a chain of per-pipeline copies and events pays the dispatch latency of a busy chip per link).

        feeder = HostFeeder(pipes)
        feeder.step(lambda p: p.replay())      # per step: upload, passes, download
        feeder.sync()                          # results of the last step: feeder.result(i)

    up / down: the streams that carry the transfers (default: new ones).  The chip has 32 hardware queues; a process
    with more live streams than that has some of them SHARE a queue, and a copy queued behind another pipeline's pass
    in the shared queue serialises the step.  With 30 or more pipeline streams pass two streams that carry nothing
    else."""

    def __init__(self, pipes, up=None, down=None):
        self.pipes = list(pipes)
        dev = self.pipes[0].dev
        self.up = up if up is not None else torch.cuda.Stream(device=dev)
        self.down = down if down is not None else torch.cuda.Stream(device=dev)
        n_in = [(p.src.x.numel(), p.tgt.x.numel()) for p in self.pipes]
        n_out = [p.wave.numel() for p in self.pipes]
        self.in_off = np.concatenate(([0], np.cumsum([a + b for a, b in n_in]))).astype(np.int64)
        self.out_off = np.concatenate(([0], np.cumsum(n_out))).astype(np.int64)
        f64 = dict(dtype=torch.float64)
        self.host_in = torch.empty(int(self.in_off[-1]), **f64).pin_memory()
        self.host_out = [torch.empty(int(self.out_off[-1]), **f64).pin_memory() for _ in range(2)]
        for i, p in enumerate(self.pipes):
            a = int(self.in_off[i])
            self.host_in[a:a + n_in[i][0]].copy_(p.src.x)
            self.host_in[a + n_in[i][0]:a + n_in[i][0] + n_in[i][1]].copy_(p.tgt.x)
        self.n_in = n_in
        self.slots = [dict(dev_in=torch.empty(int(self.in_off[-1]), device=dev, **f64),
                           dev_out=torch.empty(int(self.out_off[-1]), device=dev, **f64),
                           taken=[], drained=None) for _ in range(2)]
        self.n = 0

    def step(self, launch):
        k = self.n & 1
        self.n += 1
        sl = self.slots[k]
        for ev in sl['taken']:                   # the passes of two steps ago have copied this slot in
            self.up.wait_event(ev)
        with torch.cuda.stream(self.up):
            sl['dev_in'].copy_(self.host_in, non_blocking=True)
            arrived = torch.cuda.Event()
            arrived.record(self.up)
        sl['taken'] = []
        done = []
        for i, p in enumerate(self.pipes):
            a, (ns, nt) = int(self.in_off[i]), self.n_in[i]
            o = int(self.out_off[i])
            p.stream.wait_event(arrived)
            with torch.cuda.stream(p.stream):
                torch.mul(sl['dev_in'][a:a + ns], 1.0, out=p.src.x)
                torch.mul(sl['dev_in'][a + ns:a + ns + nt], 1.0, out=p.tgt.x)
                ev = torch.cuda.Event()
                ev.record(p.stream)
                sl['taken'].append(ev)
            launch(p)
            if sl['drained'] is not None:
                p.stream.wait_event(sl['drained'])        # the download of two steps ago has left this slot
            with torch.cuda.stream(p.stream):
                torch.mul(p.wave, 1.0, out=sl['dev_out'][o:o + p.wave.numel()])
                ev = torch.cuda.Event()
                ev.record(p.stream)
                done.append(ev)
        for ev in done:
            self.down.wait_event(ev)
        with torch.cuda.stream(self.down):
            self.host_out[k].copy_(sl['dev_out'], non_blocking=True)
            sl['drained'] = torch.cuda.Event()
            sl['drained'].record(self.down)

    def result(self, i):
        """pipeline i's waveform of the last step (pinned host memory; call sync() first)"""
        k = (self.n - 1) & 1
        return self.host_out[k][int(self.out_off[i]):int(self.out_off[i + 1])]

    def sync(self):
        for p in self.pipes:
            p.sync()
        self.up.synchronize()
        self.down.synchronize()


class BatchHostFeeder:
    """The waveforms of a PairBatchPipeline's step from pinned host memory and its synthesised waveforms back, inside the
    step: ONE upload and ONE download per step on two service streams, two staging slots each way, so the transfers of
    neighbouring steps overlap with the kernels.  The pipeline's buffers keep their addresses (its captured graph stays
    valid): the step starts with a device-to-device copy kernel of the library out of the staging block
    (kwy_copy_dev) and ends with one into it.

        feeder = BatchHostFeeder(pipe)
        feeder.step(lambda p: p.replay())
        feeder.sync()                          # results of the last step: feeder.result(k)
    """

    def __init__(self, pipe):
        self.pipe = pipe
        dev = pipe.dev
        self.up, self.down = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        n_in = [wv.x_all.numel() for wv in pipe.waves]
        n_out = [wv.wave_all.numel() for wv in pipe.waves]
        self.in_off = np.concatenate(([0], np.cumsum(n_in))).astype(np.int64)
        self.out_off = np.concatenate(([0], np.cumsum(n_out))).astype(np.int64)
        f64 = dict(dtype=torch.float64)
        self.host_in = torch.empty(int(self.in_off[-1]), **f64).pin_memory()
        self.host_out = [torch.empty(int(self.out_off[-1]), **f64).pin_memory() for _ in range(2)]
        for w, wv in enumerate(pipe.waves):
            self.host_in[int(self.in_off[w]):int(self.in_off[w + 1])].copy_(wv.x_all)
        self.slots = [dict(dev_in=torch.empty(int(self.in_off[-1]), device=dev, **f64),
                           dev_out=torch.empty(int(self.out_off[-1]), device=dev, **f64),
                           taken=None, drained=None) for _ in range(2)]
        self.n = 0

    def step(self, launch):
        p, k = self.pipe, self.n & 1
        self.n += 1
        sl = self.slots[k]
        if sl['taken'] is not None:              # the step of two steps ago has copied this slot in
            self.up.wait_event(sl['taken'])
        with torch.cuda.stream(self.up):
            sl['dev_in'].copy_(self.host_in, non_blocking=True)
            arrived = torch.cuda.Event()
            arrived.record(self.up)
        p.stream.wait_event(arrived)
        h = p.ctx.handle
        with torch.cuda.stream(p.stream):
            for w, wv in enumerate(p.waves):
                _lib.check(p.ctx, lib.kwy_copy_dev(h, _p(wv.x_all), c_vp(sl['dev_in'].data_ptr() + 8 * int(self.in_off[w])),
                                                   8 * wv.x_all.numel()))
            sl['taken'] = torch.cuda.Event()
            sl['taken'].record(p.stream)
        launch(p)
        if sl['drained'] is not None:
            p.stream.wait_event(sl['drained'])            # the download of two steps ago has left this slot
        with torch.cuda.stream(p.stream):
            for w, wv in enumerate(p.waves):
                _lib.check(p.ctx, lib.kwy_copy_dev(h, c_vp(sl['dev_out'].data_ptr() + 8 * int(self.out_off[w])), _p(wv.wave_all),
                                                   8 * wv.wave_all.numel()))
            done = torch.cuda.Event()
            done.record(p.stream)
        self.down.wait_event(done)
        with torch.cuda.stream(self.down):
            self.host_out[k].copy_(sl['dev_out'], non_blocking=True)
            sl['drained'] = torch.cuda.Event()
            sl['drained'].record(self.down)

    def result(self, k):
        """pair k's waveform of the last step (pinned host memory; call sync() first)"""
        w, i = self.pipe.where[k]
        wv = self.pipe.waves[w]
        off = int(self.out_off[w]) + int(sum(wv.ylen[:i]))
        return self.host_out[(self.n - 1) & 1][off:off + wv.ylen[i]]

    def sync(self):
        self.pipe.sync()
        self.pipe.stream.synchronize()
        self.up.synchronize()
        self.down.synchronize()


class SilenceFeeder:
    """Fresh pad spectra for every pass of a set of PairPipelines, drawn ON THE DEVICE from numpy's legacy generator
    (kwiiyatta_amd.backend.nprandom.DeviceRandomState: the same draws `pad_silence` would take from np.random, in the
    same order -- source head, source tail, target head, target tail, pair after pair), one step ahead of the
    pipelines: the generator has its own stream and two staging slots; ONE call draws the pads of all pipelines of a
    step (the serial MT19937 part is then one kernel: a chain of per-pair launches pays the dispatch latency of a busy
    chip per link), and a pass starts by copying its four blocks into the pad rows.

        feeder = SilenceFeeder(pipes, DeviceRandomState.from_global())
        feeder.step(lambda p: p.replay())
    """

    def __init__(self, pipes, rng):
        self.pipes, self.rng = list(pipes), rng
        self.scale = EPS / self.pipes[0].fs
        self.slots = [dict(blocks=[[torch.empty((PAD_LEN, p.K), dtype=torch.float64, device=p.dev) for _ in range(4)]
                                   for p in self.pipes], taken=[]) for _ in range(2)]
        self.n = 0

    def step(self, launch):
        k = self.n & 1
        self.n += 1
        sl = self.slots[k]
        for ev in sl['taken']:
            self.rng.stream.wait_event(ev)
        self.rng.abs_normal_blocks(self.scale, [b for blocks in sl['blocks'] for b in blocks])
        ready = self.rng.record_event()
        sl['taken'] = []
        for p, blocks in zip(self.pipes, sl['blocks']):
            p.stream.wait_event(ready)
            rows = p.src.silence_rows() + p.tgt.silence_rows()
            with torch.cuda.stream(p.stream):
                for dst, blk in zip(rows, blocks):
                    torch.mul(blk, 1.0, out=dst)
                ev = torch.cuda.Event()
                ev.record(p.stream)
                sl['taken'].append(ev)
            launch(p)

    def sync(self):
        for p in self.pipes:
            p.sync()
        self.rng.stream.synchronize()


def synthetic_gmm(order=24, components=64, seed=0, n_frames=16000):
    """A fixed joint GMM for the benchmark: scikit-learn's
    GaussianMixture fitted on seed-0 synthetic joint static+delta features."""
    from sklearn.mixture import GaussianMixture
    from .backend.mlpg import DELTA_WINDOWS, delta_features
    rng = np.random.default_rng(seed)
    d = order
    # smooth random trajectories with a mel-cepstrum-like decay over the coefficients
    scale = 1.0 / (1.0 + np.arange(d)) ** 0.7
    walk = np.cumsum(rng.standard_normal((n_frames, d)), axis=0) * 0.05
    src = (walk - walk.mean(0)) * scale + rng.standard_normal((n_frames, d)) * 0.02 * scale
    mix = np.eye(d) + 0.08 * rng.standard_normal((d, d))
    tgt = src @ mix + 0.05 * scale + rng.standard_normal((n_frames, d)) * 0.02 * scale
    joint = np.hstack([delta_features(src, DELTA_WINDOWS), delta_features(tgt, DELTA_WINDOWS)])
    import warnings
    gmm = GaussianMixture(n_components=components, covariance_type='full', max_iter=2, random_state=seed,
                          reg_covar=1e-4)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')   # two EM iterations on purpose: a fixed model, not a fit
        gmm.fit(joint)
    return gmm
