"""The batched entries of round 4 (include/kwy.h: kwy_fastdtw_batch_dev, kwy_synth_plan_batch_dev,
kwy_convert_mcep_batch_dev, kwy_align_*_batch_dev, kwy_gather_rows_batch_dev) and the lockstep driver on top of them
(pipeline.PairBatchPipeline).  Integer / byte work must be equal, and here the floating-point work is too: a batch runs
the same kernels as the single calls (a single call IS a batch of one), so every job's result is compared bit for bit
with the per-pair call; the per-pair calls are checked against the oracle elsewhere (test_backends_gpu.py,
test_pipeline_gpu.py).  Reference stages: /root/reference/kwiiyatta/vocoder/align.py:61-131,
vocoder/world.py:80-92, convert_voice.py:35-46."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _dtw_single(ctx, x, y, radius):
    import torch
    from kwiiyatta_amd import _lib
    from kwiiyatta_amd._lib import lib, c_vp
    dx, dy = _dev(x), _dev(y)
    path = torch.zeros((len(x) + len(y) + 2, 2), dtype=torch.int32, device='cuda')
    n = torch.zeros(1, dtype=torch.int64, device='cuda')
    dist = torch.zeros(1, dtype=torch.float64, device='cuda')
    _lib.check(ctx, lib.kwy_fastdtw_dev(ctx.handle, c_vp(dx.data_ptr()), len(x), c_vp(dy.data_ptr()), len(y), x.shape[1],
                                        radius, c_vp(dist.data_ptr()), c_vp(path.data_ptr()), c_vp(n.data_ptr())))
    ctx.sync()
    return float(dist.item()), path.cpu().numpy()[:int(n.item())]


def _series(rng, n, dim, walk=True):
    a = rng.standard_normal((n, dim))
    return np.cumsum(a, axis=0) * 0.3 if walk else a


@pytest.mark.parametrize('radius', [1, 5, 32])
def test_fastdtw_batch_equals_single_and_oracle(radius):
    """ragged batch (more pairs than one launch holds, different numbers of levels per pair, single-strip and
    multi-strip levels mixed) == per-pair device calls == oracle, path and distance exact"""
    import torch
    from oracle import oracle as ko
    from kwiiyatta_amd import _lib
    from kwiiyatta_amd._lib import lib
    rng = np.random.default_rng(100 + radius)
    shapes = [(300, 340), (64, 65), (700, 650), (33, 200), (130, 129), (1, 50), (40, 40), (257, 511), (90, 700),
              (500, 100), (65, 64), (34, 34), (35, 70), (128, 128), (200, 33), (2, 2), (420, 400), (77, 78), (600, 610)]
    dim = 5
    xs = [_series(rng, a, dim) for a, _ in shapes]
    ys = [_series(rng, b, dim) for _, b in shapes]
    ctx = _lib.Context(0)
    dxs, dys = [_dev(x) for x in xs], [_dev(y) for y in ys]
    paths = [torch.zeros((a + b + 2, 2), dtype=torch.int32, device='cuda') for a, b in shapes]
    lens = torch.zeros(len(shapes), dtype=torch.int64, device='cuda')
    dists = torch.zeros(len(shapes), dtype=torch.float64, device='cuda')
    jobs = _lib.job_array(_lib.DtwJob, [(dxs[i], shapes[i][0], dys[i], shapes[i][1], dists[i:i + 1], paths[i],
                                         lens[i:i + 1]) for i in range(len(shapes))])
    _lib.check(ctx, lib.kwy_fastdtw_batch_dev(ctx.handle, jobs, len(shapes), dim, radius))
    ctx.sync()
    for i in range(len(shapes)):
        d1, p1 = _dtw_single(ctx, xs[i], ys[i], radius)
        n = int(lens[i].item())
        pb = paths[i].cpu().numpy()[:n]
        assert np.array_equal(pb, p1), shapes[i]
        assert float(dists[i].item()) == d1, shapes[i]
        if i % 3 == 0:
            d_ref, p_ref = ko.fastdtw(xs[i], ys[i], radius=radius, dist=2)
            assert [tuple(r) for r in pb.tolist()] == p_ref and d1 == d_ref, shapes[i]


@pytest.mark.parametrize('shape', [(40, 9000), (9000, 40), (300, 1700), (2000, 330), (70, 4000)])
def test_fastdtw_dev_lopsided_pairs(shape):
    """The device entry on pairs with length ratios beyond 1:5 at radius 32 (round 3's capacity ESTIMATES overflowed
    there and the device entry returned an empty path with rc == 0; the capacities are bounds now)."""
    from oracle import oracle as ko
    from kwiiyatta_amd import _lib
    rng = np.random.default_rng(shape[0])
    x, y = _series(rng, shape[0], 4), _series(rng, shape[1], 4)
    ctx = _lib.Context(0)
    d, p = _dtw_single(ctx, x, y, 32)
    d_ref, p_ref = ko.fastdtw(x, y, radius=32, dist=2)
    assert len(p) > 0 and [tuple(r) for r in p.tolist()] == p_ref and d == d_ref


def test_fastdtw_adversarial_windows_fit_the_bounds():
    """series built so that the coarse paths run along the matrix' edges (long horizontal and vertical stretches: the
    widest windows a level can have); the device path still equals the oracle's"""
    from oracle import oracle as ko
    from kwiiyatta_amd import _lib
    ctx = _lib.Context(0)
    for tx, ty in ((900, 900), (1200, 500)):
        x = np.r_[np.zeros(tx // 2), np.linspace(0, 50, tx - tx // 2)][:, None] * np.ones((1, 3))
        y = np.r_[np.linspace(0, 50, ty // 3), np.full(ty - ty // 3, 50.0)][:, None] * np.ones((1, 3))
        d, p = _dtw_single(ctx, x, y, 32)
        d_ref, p_ref = ko.fastdtw(x, y, radius=32, dist=2)
        assert [tuple(r) for r in p.tolist()] == p_ref and d == d_ref


def test_synth_plan_batch_equals_single():
    import torch
    from kwiiyatta_amd import _lib
    from kwiiyatta_amd._lib import lib, c_vp
    from kwiiyatta_amd.synthetic import make_utterance
    fs, fft = 48000, 2048
    secs = [0.4, 1.3, 0.05, 0.9, 2.1, 0.7, 0.011, 1.0, 0.6, 0.3, 1.7, 0.2, 0.8, 0.5, 1.1, 0.45, 0.95, 0.33]
    f0s = []
    for i, sec in enumerate(secs):
        f0 = make_utterance(seed=i, fs=fs, seconds=sec)[1]
        if i == 3:
            f0 = np.zeros_like(f0)                      # unvoiced throughout
        if i == 4:
            f0[:200] = 0.0                              # leading silence
        f0s.append(f0)
    ctx = _lib.Context(0)
    ylens = [int(lib.kwy_synth_length(len(f), 5.0, fs)) for f in f0s]
    d_f0 = [_dev(f) for f in f0s]
    plans_b = [torch.zeros(int(lib.kwy_synth_plan_bytes(y)), dtype=torch.uint8, device='cuda') for y in ylens]
    plans_s = [torch.zeros_like(p) for p in plans_b]
    jobs = _lib.job_array(_lib.SynthPlanJob, [(d_f0[i], len(f0s[i]), ylens[i], plans_b[i]) for i in range(len(secs))])
    _lib.check(ctx, lib.kwy_synth_plan_batch_dev(ctx.handle, jobs, len(secs), fft, 5.0, fs))
    for i in range(len(secs)):
        _lib.check(ctx, lib.kwy_synth_plan_dev(ctx.handle, c_vp(d_f0[i].data_ptr()), len(f0s[i]), fft, 5.0, fs, ylens[i],
                                               c_vp(plans_s[i].data_ptr())))
    ctx.sync()
    # the plan holds {pulse count, tile offsets, voicing bytes, pulse indices, shifts}; beyond the pulse count the
    # index / shift arrays are unwritten in both (zero-initialised here)
    for i in range(len(secs)):
        assert torch.equal(plans_b[i], plans_s[i]), i
    # and a rendering from a batched plan equals the one-call synthesis
    K = fft // 2 + 1
    rng = np.random.default_rng(0)
    i = 1
    T = len(f0s[i])
    sp = _dev(np.abs(rng.standard_normal((T, K))) * 1e-6 + 1e-8)
    ap = _dev(np.clip(rng.random((T, K)), 0.01, 0.99))
    y1 = torch.empty(ylens[i], dtype=torch.float64, device='cuda')
    y2 = torch.empty_like(y1)
    _lib.check(ctx, lib.kwy_synth_render_dev(ctx.handle, c_vp(plans_b[i].data_ptr()), T, c_vp(sp.data_ptr()),
                                             c_vp(ap.data_ptr()), fft, 5.0, fs, 1.0, ylens[i], c_vp(y1.data_ptr())))
    _lib.check(ctx, lib.kwy_synthesize_dev(ctx.handle, c_vp(d_f0[i].data_ptr()), T, c_vp(sp.data_ptr()),
                                           c_vp(ap.data_ptr()), fft, 5.0, fs, 1.0, ylens[i], c_vp(y2.data_ptr())))
    ctx.sync()
    assert torch.equal(y1, y2)


def test_convert_mcep_batch_equals_single():
    import torch
    from kwiiyatta_amd import _lib, pipeline as pl
    from kwiiyatta_amd._lib import lib, c_vp
    order = 24
    gmm = pl.synthetic_gmm(order=order, components=8, seed=0, n_frames=4000)
    dg = pl.DeviceGMM(gmm.weights_, gmm.means_, gmm.covariances_, torch.device('cuda', 0))
    model = dg.model(diff=False)
    rng = np.random.default_rng(3)
    Ts = [1, 2, 5, 16, 17, 31, 33, 100, 257, 1031, 64, 640, 15, 48, 333, 2201, 77, 9]       # 18 > one launch
    scale = 1.0 / (1.0 + np.arange(order + 1)) ** 0.7
    mcs = [_dev(np.cumsum(rng.standard_normal((T, order + 1)), axis=0) * 0.05 * scale) for T in Ts]
    outs_b = [torch.zeros_like(m) for m in mcs]
    outs_s = [torch.zeros_like(m) for m in mcs]
    ctx = _lib.Context(0)
    jobs = _lib.job_array(_lib.ConvertJob, [(mcs[i], Ts[i], outs_b[i]) for i in range(len(Ts))])
    _lib.check(ctx, lib.kwy_convert_mcep_batch_dev(ctx.handle, jobs, len(Ts), order, dg.M, c_vp(model.data_ptr())))
    for i in range(len(Ts)):
        _lib.check(ctx, lib.kwy_convert_mcep_dev(ctx.handle, c_vp(mcs[i].data_ptr()), Ts[i], order, dg.M,
                                                 c_vp(model.data_ptr()), c_vp(outs_s[i].data_ptr())))
    ctx.sync()
    for i in range(len(Ts)):
        assert torch.equal(outs_b[i], outs_s[i]), Ts[i]
        assert torch.equal(outs_b[i][:, 0], mcs[i][:, 0])


def test_align_batches_equal_single():
    import torch
    from kwiiyatta_amd import _lib
    from kwiiyatta_amd._lib import lib, c_vp
    rng = np.random.default_rng(8)
    ctx = _lib.Context(0)
    nc = 25
    Ts = [3, 8, 9, 500, 1234, 64] * 12                     # 72 jobs: more than one launch
    mcs = [_dev(rng.standard_normal((T, nc))) for T in Ts]
    f0s = [_dev(np.where(rng.random(T) > 0.4, 120.0, 0.0)) for T in Ts]
    ob = [torch.zeros((T, nc + 1), dtype=torch.float64, device='cuda') for T in Ts]
    os_ = [torch.zeros_like(o) for o in ob]
    jobs = _lib.job_array(_lib.AlignJob, [(mcs[i], f0s[i], Ts[i], ob[i]) for i in range(len(Ts))])
    _lib.check(ctx, lib.kwy_align_features_batch_dev(ctx.handle, jobs, len(Ts), nc, 9.4, 1.636, 9.0))
    for i in range(len(Ts)):
        _lib.check(ctx, lib.kwy_align_features_dev(ctx.handle, c_vp(mcs[i].data_ptr()), Ts[i], nc, c_vp(f0s[i].data_ptr()),
                                                   9.4, 1.636, 9.0, c_vp(os_[i].data_ptr())))
    ctx.sync()
    for i in range(len(Ts)):
        assert torch.equal(ob[i], os_[i])
    # gathers
    width = 1025
    srcs = [_dev(rng.standard_normal((T, width))) for T in Ts[:6]]
    idxs = [_dev(rng.integers(-2, T + 2, size=n).astype(np.int32)) for T, n in zip(Ts[:6], (5, 1, 20, 700, 100, 64))]
    dst = [torch.zeros((len(ix), width), dtype=torch.float64, device='cuda') for ix in idxs]
    jobs = _lib.job_array(_lib.GatherJob, [(srcs[i], Ts[i], idxs[i], len(idxs[i]), dst[i]) for i in range(6)])
    _lib.check(ctx, lib.kwy_gather_rows_batch_dev(ctx.handle, jobs, 6, width))
    ctx.sync()
    for i in range(6):
        ref = srcs[i][idxs[i].long().clamp(0, Ts[i] - 1)]
        assert torch.equal(dst[i], ref)


def _pairs(n, fs=48000):
    from kwiiyatta_amd.synthetic import make_utterance
    out = []
    for i in range(n):
        sec = 0.5 + 0.13 * (i % 5)
        src = make_utterance(seed=100 + i, fs=fs, seconds=sec)
        tgt = make_utterance(seed=200 + i, fs=fs, seconds=sec, time_warp=1.0 + 0.04 * (i % 4), formant_scale=1.12)
        out.append((src, tgt))
    return out


def test_pair_batch_pipeline_equals_pair_pipelines():
    """ragged pairs in two waves, explicit pad blocks: every waveform, path and distance equals PairPipeline's bit for
    bit; a captured graph replays to the same bits"""
    import torch
    from kwiiyatta_amd import pipeline as pl
    fs = 48000
    pairs = _pairs(7, fs)
    gmm = pl.synthetic_gmm(order=24, components=8, seed=0, n_frames=4000)
    dg = pl.DeviceGMM(gmm.weights_, gmm.means_, gmm.covariances_, torch.device('cuda', 0))
    K = 1025
    rng = np.random.default_rng(1)
    silence = [np.abs(rng.standard_normal((pl.PAD_LEN, K))) * pl.EPS / fs for _ in range(4 * len(pairs))]
    b = pl.PairBatchPipeline(0, fs, pairs, dg, waves=2, silence=silence, fused_mcep=False)   # the reference's two stages
    assert len(b.waves) == 2
    b.run()
    b.sync()
    waves = [b.wave(k).clone() for k in range(len(pairs))]
    for k, (src, tgt) in enumerate(pairs):
        p = pl.PairPipeline(0, fs, src, tgt, dg, silence=silence[4 * k:4 * k + 4])
        p.run()
        p.sync()
        path, n, dist = b.path(k)
        assert int(n.item()) == int(p.path_len.item()) and torch.equal(path[:int(n.item())], p.path[:int(n.item())])
        assert float(dist.item()) == float(p.dist.item())
        assert torch.equal(waves[k], p.wave), k
    b.capture()
    for _ in range(2):
        b.replay()
    b.sync()
    for k in range(len(pairs)):
        assert torch.equal(b.wave(k), waves[k])


def test_pair_batch_pipeline_draws_its_pads_like_align():
    """with a DeviceRandomState the step draws fresh pads for all pairs in the reference's order: the generator state
    afterwards equals numpy's after the same draws, the pads equal numpy's (to the ulp of log), and the waveforms equal
    those of pair pipelines fed the device-drawn pads"""
    import torch
    from kwiiyatta_amd import pipeline as pl
    from kwiiyatta_amd.backend.nprandom import DeviceRandomState
    fs, K = 48000, 1025
    pairs = _pairs(5, fs)
    gmm = pl.synthetic_gmm(order=24, components=8, seed=0, n_frames=4000)
    dg = pl.DeviceGMM(gmm.weights_, gmm.means_, gmm.covariances_, torch.device('cuda', 0))
    rs = DeviceRandomState.from_seed(4242)
    b = pl.PairBatchPipeline(0, fs, pairs, dg, waves=2, rng=rs, fused_mcep=False)
    b.run()
    b.sync()
    rs.sync()
    ref = np.random.RandomState(4242)
    pads_ref = [np.abs(ref.normal(0, pl.EPS / fs, (pl.PAD_LEN, K))) for _ in range(4 * len(pairs))]
    pads = [blk.cpu().numpy() for blk in b.pad_rows]
    for got, exp in zip(pads, pads_ref):
        assert np.abs(got - exp).max() <= 4e-16 * exp.max()
    st_dev, st_np = rs.get_state(), ref.get_state()
    assert np.array_equal(st_dev[1], st_np[1]) and st_dev[2:] == tuple(st_np[2:])
    waves = [b.wave(k).clone() for k in range(len(pairs))]
    for k, (src, tgt) in enumerate(pairs):
        p = pl.PairPipeline(0, fs, src, tgt, dg, silence=pads[4 * k:4 * k + 4])
        p.run()
        p.sync()
        assert torch.equal(waves[k], p.wave), k
    # a captured step draws again at every replay: run, capture()'s plain pass and one replay are three passes of
    # 4 blocks per pair each (recording the graph executes nothing)
    b.capture()
    b.replay()
    b.sync()
    rs.sync()
    ref2 = np.random.RandomState(4242)
    for _ in range(3 * 4 * len(pairs)):
        ref2.normal(0, 1.0, (pl.PAD_LEN, K))
    st_dev, st_np = rs.get_state(), ref2.get_state()
    assert np.array_equal(st_dev[1], st_np[1]) and st_dev[2:] == tuple(st_np[2:])


def test_cheaptrick_mcep_entry_and_fused_lockstep_step():
    """kwy_cheaptrick_mcep_batch_dev (CheapTrick's liftered cepstrum through pysptk's frequency transform, no envelope
    row in between) against the two calls it replaces: the oracle's sp2mc(cheaptrick(x) / fs) within 1e-12 of the
    coefficients' scale, more utterances than one launch holds; and the lockstep step built on it against the step
    that runs the two stages: same FastDTW paths, mel-cepstra within 1e-12, waveforms within 1e-9."""
    import torch
    from oracle import oracle as ko
    from kwiiyatta_amd import _lib, pipeline as pl
    from kwiiyatta_amd._lib import lib
    from kwiiyatta_amd.backend import sptk
    from kwiiyatta_amd.synthetic import make_utterance
    fs, order = 48000, 24
    alpha = sptk.mcepalpha(fs)
    utts = [make_utterance(seed=40 + i, fs=fs, seconds=0.3 + 0.07 * i, f0_base=100.0 + 9 * i) for i in range(18)]
    ctx = _lib.Context(0)
    dev = [tuple(_dev(a) for a in u) for u in utts]
    outs = [torch.full((len(u[1]), order + 1), float('nan'), dtype=torch.float64, device='cuda') for u in utts]
    arr = _lib.utterance_array([(d[0], d[2], d[1], o) for d, o in zip(dev, outs)])
    torch.cuda.synchronize()          # (torch filled `outs` on its stream; the library runs on the context's)
    _lib.check(ctx, lib.kwy_cheaptrick_mcep_batch_dev(ctx.handle, arr, len(utts), fs, -0.15, 71.0, 2048, float(fs), order, alpha))
    ctx.sync()
    for i in (0, 5, 17):
        x, f0, t = utts[i]
        ref = ko.sp2mc(np.ascontiguousarray(ko.cheaptrick(x, f0, t, fs) / fs), order, alpha)
        got = outs[i].cpu().numpy()
        assert got.shape == ref.shape and np.isfinite(got).all()
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max(), (i, np.abs(got - ref).max())
    pairs = _pairs(5, fs)
    gmm = pl.synthetic_gmm(order=24, components=8, seed=0, n_frames=4000)
    dg = pl.DeviceGMM(gmm.weights_, gmm.means_, gmm.covariances_, torch.device('cuda', 0))
    rng = np.random.default_rng(2)
    silence = [np.abs(rng.standard_normal((pl.PAD_LEN, 1025))) * pl.EPS / fs for _ in range(4 * len(pairs))]
    a = pl.PairBatchPipeline(0, fs, pairs, dg, waves=2, silence=silence)                     # fused (default)
    b = pl.PairBatchPipeline(0, fs, pairs, dg, waves=2, silence=silence, fused_mcep=False)
    for p in (a, b):
        p.run()
        p.sync()
    for wa, wb in zip(a.waves, b.waves):
        ma, mb = wa.mc_pad.cpu().numpy(), wb.mc_pad.cpu().numpy()
        assert np.abs(ma - mb).max() <= 1e-12 * np.abs(mb).max()
    for k in range(len(pairs)):
        pa, na, da = a.path(k)
        pb, nb, db = b.path(k)
        assert int(na.item()) == int(nb.item()) and torch.equal(pa[:int(na.item())], pb[:int(nb.item())])
        assert float((a.wave(k) - b.wave(k)).abs().max()) <= 1e-9
    first = [a.wave(k).clone() for k in range(len(pairs))]
    a.capture()
    a.replay()
    a.sync()
    for k in range(len(pairs)):
        assert torch.equal(a.wave(k), first[k])
