"""The HBM-resident pair pipeline (kwiiyatta_amd.pipeline) checked stage by
stage against the CPU oracle and the host-side reference logic, on a short
synthetic source/target pair (BASELINE config 3 in miniature)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def pair():
    from kwiiyatta_amd.synthetic import make_utterance
    fs = 48000
    src = make_utterance(seed=1234, fs=fs, seconds=1.5)
    tgt = make_utterance(seed=4321, fs=fs, seconds=1.5, time_warp=1.1, formant_scale=1.12)
    return fs, src, tgt


def test_pair_pipeline_stages(pair):
    import torch
    from oracle import oracle as ko
    from kwiiyatta_amd import pipeline as pl
    from kwiiyatta_amd.vocoder.align import project_path_iter
    fs, src, tgt = pair
    gmm = pl.synthetic_gmm(order=24, components=8, seed=0, n_frames=4000)
    dg = pl.DeviceGMM(gmm.weights_, gmm.means_, gmm.covariances_, torch.device('cuda', 0))
    p = pl.PairPipeline(0, fs, src, tgt, dg, keep_aligned_spectrum=True)
    p.run()
    p.sync()
    P = pl.PAD_LEN
    alpha = ko.mcepalpha(fs)
    for side, (x, f0, t) in ((p.src, src), (p.tgt, tgt)):
        sp_pad = side.sp_pad.cpu().numpy()
        ap_pad = side.ap_pad.cpu().numpy()
        sp_ref = ko.cheaptrick(x, f0, t, fs) / fs
        assert np.abs(sp_pad[P:P + len(f0)] - sp_ref).max() <= 1e-8 * sp_ref.max()
        assert np.abs(ap_pad[P:P + len(f0)] - ko.d4c(x, f0, t, fs)).max() <= 1e-4
        assert (ap_pad[:P] == 1 - 1e-12).all() and (ap_pad[P + len(f0):] == 1 - 1e-12).all()
        sil = np.r_[sp_pad[:P], sp_pad[P + len(f0):]]
        assert (sil > 0).all() and sil.max() < 10 * 2.2e-16 / fs
        mc_ref = ko.sp2mc(sp_pad, 24, alpha)
        mc = side.mc_pad.cpu().numpy()
        assert np.abs(mc - mc_ref).max() <= 1e-11 * np.abs(mc_ref).max()
        # make_feature(vuv='f0', power='binalize', power_pivot='max')
        feat = side.feat.cpu().numpy()
        f0_pad = np.r_[np.zeros(P), f0, np.zeros(P)]
        exp = np.hstack((np.zeros((len(mc), 2)), mc[:, 1:]))
        exp[:, 0][mc[:, 0] >= mc[:, 0].max() - 1.636] = 9.4
        exp[:, 1][f0_pad > 0] = 9.0
        assert np.array_equal(feat, exp)
    # FastDTW on the device features: bit-exact path
    fs_, ft_ = p.src.feat.cpu().numpy(), p.tgt.feat.cpu().numpy()
    d_ref, path_ref = ko.fastdtw(fs_, ft_, radius=32, dist=2)
    n = int(p.path_len.item())
    path = [tuple(r) for r in p.path.cpu().numpy()[:n].tolist()]
    assert path == path_ref and p.dist.item() == d_ref
    # project_path_iter and the row gathers
    idx_ref = list(project_path_iter(np.array(path_ref), trim=True, trim_len=P))
    assert int(p.n_idx.item()) == len(idx_ref) == p.tgt.T
    idx = p.idx.cpu().numpy()[:p.tgt.T]
    assert idx.tolist() == idx_ref
    assert np.array_equal(p.sp_al.cpu().numpy(), p.src.sp_pad.cpu().numpy()[idx])
    assert np.array_equal(p.ap_al.cpu().numpy(), p.src.ap_pad.cpu().numpy()[idx])
    mc_al = p.mc_al.cpu().numpy()
    assert np.array_equal(mc_al, p.src.mc_pad.cpu().numpy()[idx])
    # conversion, spectrum, waveform
    y_ref = ko.gmm_mlpg(np.ascontiguousarray(mc_al[:, 1:]), gmm.weights_, gmm.means_, gmm.covariances_)
    mc_conv = p.mc_conv.cpu().numpy()
    assert np.array_equal(mc_conv[:, 0], mc_al[:, 0])
    assert np.abs(mc_conv[:, 1:] - y_ref).max() <= 1e-9 * max(np.abs(y_ref).max(), 1)
    sp_conv = p.sp_conv.cpu().numpy()
    assert np.max(np.abs(sp_conv / ko.mc2sp(mc_conv, alpha, 2048) - 1)) <= 1e-10
    wave_ref = ko.synthesize(tgt[1], np.ascontiguousarray(sp_conv * fs), p.ap_al.cpu().numpy(), fs, 5.0)
    wave = p.wave.cpu().numpy()
    assert len(wave) == len(wave_ref)
    assert np.sqrt(np.mean((wave - wave_ref) ** 2)) <= 1e-9


def test_streams_are_independent(pair):
    """Several pipelines enqueued back to back on their own streams give the
    same results as each alone (utterance-per-stream sharding)."""
    import torch
    from kwiiyatta_amd import pipeline as pl
    fs, src, tgt = pair
    ups = [pl.UtterancePipeline(0, fs, u) for u in (src, tgt, src)]
    for _ in range(2):
        for u in ups:
            u.run()
    for u in ups:
        u.sync()
    torch.cuda.synchronize()
    assert torch.equal(ups[0].sp, ups[2].sp) and torch.equal(ups[0].ap, ups[2].ap)
    assert torch.equal(ups[0].wave, ups[2].wave)      # the overlap-add has a fixed summation order
    assert ups[1].wave.shape != ups[0].wave.shape


@pytest.mark.parametrize('kind', ['unit', 'unit_trim0', 'gaps', 'backsteps', 'late_start', 'long'])
def test_align_project_matches_generator(kind):
    """kwy_align_project_dev against project_path_iter: DTW-shaped paths (the all-thread case) and
    paths with jumps / repeated rows (the literal walk)."""
    import torch
    from kwiiyatta_amd import _lib
    from kwiiyatta_amd.vocoder.align import project_path_iter
    rng = np.random.default_rng(5)

    def dtw_like(n):
        steps = rng.integers(0, 3, n)                    # 0: x+1, 1: y+1, 2: both
        x = np.r_[0, np.cumsum(steps != 1)]
        y = np.r_[0, np.cumsum(steps != 0)]
        return np.stack([x, y], 1)

    trim = 7
    if kind == 'unit':
        path = dtw_like(400)
    elif kind == 'unit_trim0':
        path, trim = dtw_like(300), 0
    elif kind == 'gaps':
        path = dtw_like(400)
        path = np.delete(path, [50, 51, 52, 200, 201, 350], axis=0)
    elif kind == 'backsteps':
        path = dtw_like(300)
        path[120:125, 1] = path[119, 1] - 2
    elif kind == 'late_start':
        path = dtw_like(300) + np.array([3, 4])
    else:
        path = dtw_like(9000)
    ref = list(project_path_iter(path, trim=trim > 0, trim_len=trim))
    dev = torch.device('cuda', 0)
    ctx = _lib.default_context()
    dpath = torch.from_numpy(np.ascontiguousarray(path.astype(np.int32))).to(dev)
    dlen = torch.tensor([len(path)], dtype=torch.int64, device=dev)
    cap = int(path[-1, 1]) + 8
    idx = torch.full((cap,), -7, dtype=torch.int32, device=dev)
    n_out = torch.zeros(1, dtype=torch.int64, device=dev)
    rc = _lib.lib.kwy_align_project_dev(ctx.handle, _lib.c_vp(dpath.data_ptr()), _lib.c_vp(dlen.data_ptr()), trim,
                                        _lib.c_vp(idx.data_ptr()), cap, _lib.c_vp(n_out.data_ptr()))
    _lib.check(ctx, rc)
    torch.cuda.synchronize()
    n = int(n_out.item())
    assert n == len(ref)
    assert idx[:n].cpu().tolist() == ref


def test_graph_replay_matches_plain_pass(pair):
    """A pass captured as a HIP graph and replayed (bench.py's default) does the work of the plain pass:
    same path, same converted features, same waveform bit for bit."""
    import torch
    from kwiiyatta_amd import pipeline as pl
    fs, src, tgt = pair
    gmm = pl.synthetic_gmm(order=24, components=8, seed=0, n_frames=4000)
    dg = pl.DeviceGMM(gmm.weights_, gmm.means_, gmm.covariances_, torch.device('cuda', 0))
    p = pl.PairPipeline(0, fs, src, tgt, dg)
    p.run()
    p.sync()
    p.run()
    p.sync()
    ref = {k: getattr(p, k).clone() for k in ('path', 'path_len', 'idx', 'mc_conv', 'sp_conv', 'wave')}
    p.capture()
    for k in ('path', 'idx', 'mc_conv', 'sp_conv', 'wave'):
        getattr(p, k).zero_()
    for _ in range(2):
        p.replay()
    p.sync()
    n = int(ref['path_len'].item())
    assert int(p.path_len.item()) == n
    assert torch.equal(p.path[:n], ref['path'][:n]) and torch.equal(p.idx, ref['idx'])
    assert torch.equal(p.mc_conv, ref['mc_conv']) and torch.equal(p.sp_conv, ref['sp_conv'])
    assert torch.equal(p.wave, ref['wave'])


def test_side_stream_pass_matches_single_stream(pair):
    """D4C on a second stream beside the alignment (the latency option) changes the order of execution only:
    plain and captured passes give the single-stream results bit for bit."""
    import numpy as np
    import torch
    from kwiiyatta_amd import pipeline as pl
    fs, src, tgt = pair
    gmm = pl.synthetic_gmm(order=24, components=8, seed=0, n_frames=4000)
    dg = pl.DeviceGMM(gmm.weights_, gmm.means_, gmm.covariances_, torch.device('cuda', 0))
    silence = [pl.draw_silence(fs, 1025) for _ in range(4)]
    one = pl.PairPipeline(0, fs, src, tgt, dg, silence=silence)
    two = pl.PairPipeline(0, fs, src, tgt, dg, silence=silence, side_stream=True)
    assert len(two.contexts()) == 2 and len(one.contexts()) == 1
    one.run(); one.sync()
    two.run(); two.sync()
    keys = ('path', 'idx', 'ap_al', 'mc_conv', 'sp_conv', 'wave')
    for k in keys:
        assert torch.equal(getattr(one, k), getattr(two, k)), k
    assert torch.equal(one.tgt.ap, two.tgt.ap)
    two.capture()
    for k in keys:
        getattr(two, k).zero_()
    two.src.ap.zero_()
    for _ in range(2):
        two.replay()
    two.sync()
    for k in keys:
        assert torch.equal(getattr(one, k), getattr(two, k)), k


def test_pipeline_matches_api_path_under_seed():
    """The silence pads are drawn on the host from numpy's global generator in the reference's order, so the
    HBM-resident pipeline and the package's Python API (`kwiiyatta.align`, the path the reference's CLIs take)
    see the same padded features under `np.random.seed`: same DTW path, same aligned mel-cepstra, bit for bit."""
    import torch
    import kwiiyatta_amd as kwiiyatta
    from kwiiyatta_amd import pipeline as pl
    from kwiiyatta_amd.backend import world
    from kwiiyatta_amd.synthetic import make_utterance
    fs = 16000
    sides = []
    for seed, warp, form in ((31, 1.0, 1.0), (32, 1.1, 1.12)):
        x, _, _ = make_utterance(seed=seed, fs=fs, seconds=1.3, time_warp=warp, formant_scale=form)
        f0, t = world.dio(x, fs, frame_period=5)
        sides.append((x, world.stonemask(x, f0, t, fs), t))
    np.random.seed(7)
    a, b = (kwiiyatta.Analyzer(kwiiyatta.Wavdata(fs, s[0])) for s in sides)
    aligned = kwiiyatta.align(a, b)
    gmm = pl.synthetic_gmm(order=24, components=4, seed=0, n_frames=3000)
    dg = pl.DeviceGMM(gmm.weights_, gmm.means_, gmm.covariances_, torch.device('cuda', 0))
    np.random.seed(7)
    p = pl.PairPipeline(0, fs, sides[0], sides[1], dg, keep_aligned_spectrum=True)
    p.run()
    p.sync()
    assert np.array_equal(p.mc_al.cpu().numpy(), aligned.mel_cepstrum.data)
    assert np.array_equal(p.sp_al.cpu().numpy(), aligned.spectrum_envelope)
    assert np.array_equal(p.ap_al.cpu().numpy(), aligned.aperiodicity)


def test_config4_batch_of_256_utterances():
    """BASELINE config 4 on one GPU: 256 synthetic 48 kHz utterances (seeds 0 ... 255) analysed and resynthesised
    through a fixed pool of 16 streams -- 16x more utterances than streams, every stream reusing one pipeline and,
    from its second utterance on, one captured graph.  All outputs finite, a second pass bit-identical, and a sample
    of utterances against the CPU oracle's own analyse -> synthesise chain (north star: 1e-4 RMS)."""
    import torch
    from oracle import oracle as ko
    from kwiiyatta_amd import corpus as cp
    from kwiiyatta_amd.synthetic import make_utterance
    fs, n_utt = 48000, 256
    utts = [make_utterance(seed=s, fs=fs, seconds=0.5, f0_base=110.0 + (s % 7) * 15.0) for s in range(n_utt)]
    dev = torch.device('cuda', 0)
    resident = [tuple(torch.from_numpy(a).to(dev) for a in u) for u in utts]     # inputs live in HBM
    pool = cp.StreamPool(0, 16)
    waves, frames = cp.resynthesize_batch(resident, fs, pool=pool)
    assert frames == sum(len(u[1]) for u in utts) == 256 * 101
    first = [w.cpu().numpy().copy() for w in waves]
    assert all(np.isfinite(w).all() and np.abs(w).max() > 1e-3 for w in first)
    waves2, _ = cp.resynthesize_batch(resident, fs, pool=pool)
    assert all(np.array_equal(a, b.cpu().numpy()) for a, b in zip(first, waves2))
    worst = 0.0
    for i in (0, 17, 100, 255):
        x, f0, t = utts[i]
        sp = ko.cheaptrick(x, f0, t, fs)
        ap = ko.d4c(x, f0, t, fs)
        ref = ko.synthesize(f0, sp, ap, fs, 5.0)
        assert ref.shape == first[i].shape
        worst = max(worst, float(np.sqrt(np.mean((first[i] - ref) ** 2))))
    print(f'config 4: 256 utterances on 16 streams, worst RMS against the all-oracle chain {worst:.3e}')
    assert worst <= 1e-4


def test_host_and_silence_feeders(pair):
    """HostFeeder (waveforms through pinned memory, one block each way per step, double-buffered) returns the
    waveforms of plain passes; SilenceFeeder fills the pad rows with numpy's own draws in the reference's order
    (pair after pair: source head, source tail, target head, target tail), a step ahead of the passes."""
    import torch
    from kwiiyatta_amd import pipeline as pl
    from kwiiyatta_amd.backend.nprandom import DeviceRandomState
    fs, src, tgt = pair
    gmm = pl.synthetic_gmm(order=24, components=4, seed=0, n_frames=3000)
    dg = pl.DeviceGMM(gmm.weights_, gmm.means_, gmm.covariances_, torch.device('cuda', 0))
    sil = [[pl.draw_silence(fs, 1025) for _ in range(4)] for _ in range(3)]
    pipes = [pl.PairPipeline(0, fs, src if i != 1 else tgt, tgt if i != 1 else src, dg, silence=sil[i]) for i in range(3)]
    want = []
    for p in pipes:
        p.run()
        p.sync()
        want.append(p.wave.cpu().numpy().copy())
        p.capture()
    feeder = pl.HostFeeder(pipes)
    for _ in range(3):
        feeder.step(lambda p: p.replay())
    feeder.sync()
    for i, w in enumerate(want):
        assert np.array_equal(feeder.result(i).numpy(), w)
    # pads drawn on the device, three steps: the rows of the last step are numpy's draws number 2 * 12 ... 3 * 12 - 1
    ref = np.random.RandomState(123)
    sf = pl.SilenceFeeder(pipes, DeviceRandomState.from_seed(123))
    for _ in range(3):
        sf.step(lambda p: p.replay())
        blocks = [np.abs(ref.normal(0, pl.EPS / fs, (pl.PAD_LEN, 1025))) for _ in range(12)]
    sf.sync()
    for i, p in enumerate(pipes):
        rows = p.src.silence_rows() + p.tgt.silence_rows()
        for r, b in zip(rows, blocks[4 * i:4 * i + 4]):
            assert np.abs(r.cpu().numpy() / b - 1).max() <= 1e-15
        assert np.isfinite(p.wave.cpu().numpy()).all()
