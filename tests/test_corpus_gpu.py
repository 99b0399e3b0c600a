"""BASELINE config 5 in miniature: parallel corpus -> training matrix in HBM -> converter fit -> batch
conversion (kwiiyatta_amd.corpus), against the Python API path of the package (the reference's dataset chain,
kwiiyatta/config.py:83-104 + converter/dataset.py:61-77) under the same numpy seed, against the CPU oracle's
restatement of the same chain, and against scikit-learn for the fit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FS = 16000
ORDER = 24


@pytest.fixture(scope='module')
def corpus():
    """four short synthetic pairs with f0 tracks from the package's own DIO + StoneMask (what analyze_wav runs);
    one source carries leading / trailing digital silence so that TrimmedDataset has something to trim"""
    from kwiiyatta_amd.backend import world
    from kwiiyatta_amd.synthetic import make_utterance
    out = []
    for k in range(4):
        pair = []
        for seed, warp, form in ((100 + k, 1.0, 1.0), (200 + k, 1.1, 1.12)):
            x, _, _ = make_utterance(seed=seed, fs=FS, seconds=1.1 + 0.1 * k, time_warp=warp, formant_scale=form)
            if k == 1 and warp == 1.0:
                x = np.ascontiguousarray(np.r_[np.zeros(1200), x, np.zeros(2400)])
            f0, t = world.dio(x, FS, frame_period=5)
            f0 = world.stonemask(x, f0, t, FS)
            pair.append((x, f0, t))
        out.append(tuple(pair))
    return out


def api_matrix(kwiiyatta, corpus):
    """the reference's chain through the package's Python API"""
    from kwiiyatta_amd.converter import (DeltaFeatureDataset, MelCepstrumDataset, align_dataset,
                                         make_dataset_to_array)
    pairs = {f'{k:02}': (kwiiyatta.Analyzer(kwiiyatta.Wavdata(FS, s[0])), kwiiyatta.Analyzer(kwiiyatta.Wavdata(FS, t[0])))
             for k, (s, t) in enumerate(corpus)}
    ds = DeltaFeatureDataset(MelCepstrumDataset(align_dataset(pairs)))
    return make_dataset_to_array(ds, sorted(pairs.keys()))


def test_training_matrix_equals_api_path(corpus):
    import kwiiyatta_amd as kwiiyatta
    from kwiiyatta_amd import corpus as cp
    np.random.seed(0)
    want = api_matrix(kwiiyatta, corpus)
    np.random.seed(0)
    X, frames = cp.build_training_matrix(corpus, FS, streams=3)
    got = X.cpu().numpy()
    assert frames == sum(len(s[1]) for s, _ in corpus)
    assert got.shape == want.shape and got.shape[1] == 6 * ORDER and got.shape[0] > 300
    assert np.array_equal(got, want)


def test_training_matrix_stages_vs_oracle(corpus):
    """one pair, stage by stage against the oracle: trim length, voicing, DTW features, the strict / cut path,
    delta features and the joint rows"""
    from oracle import oracle as ko
    from kwiiyatta_amd import corpus as cp
    from kwiiyatta_amd.pipeline import PAD_LEN as P
    src, tgt = corpus[1]
    np.random.seed(3)
    sil = [cp.draw_silence(FS, 513) for _ in range(4)]
    p = cp.TrainPair(0, FS, src, tgt, silence=sil)
    p.analyse()
    p.align()
    rows = p.rows().cpu().numpy()
    alpha = ko.mcepalpha(FS)
    feats, mcs = [], []
    for side, (x, f0, t), (head, tail) in ((p.src, src, sil[:2]), (p.tgt, tgt, sil[2:])):
        sp = ko.cheaptrick(x, f0, t, FS) / FS
        ap = ko.d4c(x, f0, t, FS)
        s = np.abs(sp).sum(1)
        s[s < 1e-7] = 0
        n = len(np.trim_zeros(s))
        assert side.n == n and (side is not p.src or n < len(f0))         # the padded source is trimmed
        sp_pad = np.vstack((head, sp[:n], tail))
        ap_pad = np.vstack((np.full((P, 513), 1 - 1e-12), ap[:n], np.full((P, 513), 1 - 1e-12)))
        f0_pad = np.r_[np.zeros(P), f0[:n], np.zeros(P)]
        voiced = (f0_pad >= FS / ((513 - 1) / 2) + 1.0) & (ap_pad[:, 0] <= 0.999)
        got_v = side.voiced.cpu().numpy()[:n + 2 * P] > 0
        # D4C differs by <= 1e-4 between oracle and GPU: a frame sitting on the 0.999 edge may flip
        assert (got_v != voiced).sum() <= 2
        mc = ko.sp2mc(sp_pad, ORDER, alpha)
        mc_g = side.mc_pad.cpu().numpy()
        assert np.abs(mc_g - mc).max() <= 1e-9 * np.abs(mc).max()
        feat = np.hstack((np.zeros((len(mc_g), 2)), mc_g[:, 1:]))
        feat[:, 0][mc_g[:, 0] >= mc_g[:, 0].max() - 1.636] = 9.4
        feat[:, 1][got_v] = 9.0
        assert np.array_equal(side.feat.cpu().numpy(), feat)
        feats.append(feat)
        mcs.append(mc_g)
    # FastDTW (bit-exact on identical features), strict filter, cut: restated from align.py:61-96, 134-146
    fx, fy = feats
    _, path = ko.fastdtw(fx, fy, radius=32, dist=2)
    n_path = int(p.path_len.item())
    assert [tuple(r) for r in p.path.cpu().numpy()[:n_path].tolist()] == path

    def check(x, y):
        if (fx[x, 0] > 0) ^ (fy[y, 0] > 0):
            return False
        if (fx[x, 1] > 0) ^ (fy[y, 0] > 0):
            return False
        return True
    strict = np.array([path[0]] + [c for c in path[1:-1] if check(*c)] + [path[-1]]).T
    begin = np.argmax((strict[0] >= P) & (strict[1] >= P))
    end = np.argmax((strict[0] >= len(fx) - P) & (strict[1] >= len(fy) - P))
    sel = strict[:, begin:end]
    n_sel = int(p.n_sel.item())
    assert n_sel == sel.shape[1] and 0 < n_sel < len(path)
    assert np.array_equal(p.idx_x.cpu().numpy()[:n_sel], sel[0]) and np.array_equal(p.idx_y.cpu().numpy()[:n_sel], sel[1])
    xd = ko.delta_features(mcs[0][sel[0]][:, 1:], ko.DELTA_WINDOWS)
    yd = ko.delta_features(mcs[1][sel[1]][:, 1:], ko.DELTA_WINDOWS)
    joint = np.hstack((xd, yd))
    joint = joint[np.abs(joint).sum(1) >= 1e-7]
    assert rows.shape == joint.shape
    assert np.array_equal(rows, joint)


def test_align_even_edge_paths():
    """kwy_align_even_dev on hand-made paths: a one-cell path (yielded twice by the reference's chain), nothing
    reaching the un-padded stretch (argmax of an all-false array is 0), strict off, pad_len 0"""
    import torch
    from kwiiyatta_amd import _lib
    ctx = _lib.default_context()
    dev = torch.device('cuda', 0)

    def run(path, fx, fy, strict, pad, Tx, Ty):
        path = np.asarray(path, dtype=np.int32).reshape(-1, 2)
        cap = Tx + Ty + 2
        d = dict(device=dev)
        tp = torch.from_numpy(path).to(dev)
        ln = torch.tensor([len(path)], dtype=torch.int64, **d)
        ix, iy = torch.full((cap,), -5, dtype=torch.int32, **d), torch.full((cap,), -5, dtype=torch.int32, **d)
        n = torch.zeros(1, dtype=torch.int64, **d)
        tfx, tfy = torch.from_numpy(fx).to(dev), torch.from_numpy(fy).to(dev)
        _lib.check(ctx, _lib.lib.kwy_align_even_dev(ctx.handle, _lib.c_vp(tp.data_ptr()), _lib.c_vp(ln.data_ptr()),
                                                    _lib.c_vp(tfx.data_ptr()), _lib.c_vp(tfy.data_ptr()), fx.shape[1],
                                                    int(strict), 1, 1, Tx, Ty, pad, _lib.c_vp(ix.data_ptr()),
                                                    _lib.c_vp(iy.data_ptr()), cap, _lib.c_vp(n.data_ptr())))
        torch.cuda.synchronize()
        k = int(n.item())
        return ix[:k].cpu().tolist(), iy[:k].cpu().tolist()

    def ref(path, fx, fy, strict, pad, Tx, Ty):
        def check(x, y):
            return not ((fx[x, 0] > 0) ^ (fy[y, 0] > 0)) and not ((fx[x, 1] > 0) ^ (fy[y, 0] > 0))
        if strict:
            flat = list(path[0]) + [v for c in path[1:-1] if check(*c) for v in c] + list(path[-1])
            path = np.array(flat).reshape(-1, 2)
        pt = np.array(path).T
        if pad:
            b = np.argmax((pt[0] >= pad) & (pt[1] >= pad))
            e = np.argmax((pt[0] >= Tx - pad) & (pt[1] >= Ty - pad))
            pt = pt[:, b:e]
        return pt[0].tolist(), pt[1].tolist()

    rng = np.random.default_rng(0)
    Tx, Ty = 40, 50
    fx, fy = rng.standard_normal((Tx, 4)), rng.standard_normal((Ty, 4))
    diag = [(min(i, Tx - 1), min(i * Ty // Tx, Ty - 1)) for i in range(Tx)] + [(Tx - 1, Ty - 1)]
    for path, strict, pad in ((diag, True, 5), (diag, False, 5), (diag, True, 0), ([(0, 0)], True, 0),
                              ([(0, 0)], True, 3), ([(0, 0), (1, 1), (2, 2)], True, 10), (diag[:7], False, 8)):
        assert run(path, fx, fy, strict, pad, Tx, Ty) == ref(path, fx, fy, strict, pad, Tx, Ty), (path[:3], strict, pad)


def test_fit_and_convert(corpus):
    """fit on the device-resident matrix == scikit-learn on its host copy; ConvertPipeline == the API's pieces"""
    import torch
    from sklearn.mixture import GaussianMixture
    from kwiiyatta_amd import corpus as cp
    from kwiiyatta_amd.backend import mlpg, sptk, world
    np.random.seed(1)
    X, _ = cp.build_training_matrix(corpus, FS, streams=4)
    g = cp.fit_converter(X, components=4, seed=0, max_iter=20)
    ref = GaussianMixture(n_components=4, covariance_type='full', max_iter=20, random_state=0).fit(X.cpu().numpy())
    assert g.n_iter_ == ref.n_iter_
    assert np.allclose(g.weights_, ref.weights_, rtol=1e-6, atol=1e-10)
    assert np.allclose(g.means_, ref.means_, rtol=1e-6, atol=1e-8)
    assert np.allclose(g.covariances_, ref.covariances_, rtol=1e-5, atol=1e-9)
    sources = [s for s, _ in corpus[:3]]
    waves = cp.convert_batch(sources, FS, g, streams=2)
    alpha = sptk.mcepalpha(FS)
    for (x, f0, t), w in zip(sources, waves):
        sp = world.cheaptrick(x, f0, t, FS)
        ap = world.d4c(x, f0, t, FS)
        mc = sptk.sp2mc(sp / FS, ORDER, alpha)
        y = mlpg.MLPG(g, windows=mlpg.DELTA_WINDOWS, diff=False).transform(mlpg.delta_features(mc[:, 1:], mlpg.DELTA_WINDOWS))
        sp_conv = sptk.mc2sp(np.hstack((mc[:, :1], y)), alpha, 1024)
        want = world.synthesize(f0, np.ascontiguousarray(sp_conv * FS), ap, FS, 5.0)
        got = w.cpu().numpy()
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())


def test_convert_batch_shapes_share_one_arena(corpus):
    """short, long, short again ... on ONE stream: the long utterance makes the stream's scratch arena grow (and
    move) after the short shape's pass was captured as a HIP graph; the stale graph must be captured again, not
    replayed against freed memory (round-2 review).  Also: the graphed passes equal the plain ones bit for bit, and
    a stream keeps at most `shapes_per_stream` pipelines."""
    import torch
    from kwiiyatta_amd import corpus as cp
    from kwiiyatta_amd import pipeline as pl
    from kwiiyatta_amd.synthetic import make_utterance
    g = pl.synthetic_gmm(order=ORDER, components=4, seed=0, n_frames=3000)

    def utt(seed, seconds):
        x, f0, t = make_utterance(seed=seed, fs=FS, seconds=seconds)
        return x, f0, t
    short, long_, mid = utt(1, 0.6), utt(2, 3.0), utt(3, 1.3)
    short2 = (np.ascontiguousarray(short[0][::-1]), short[1], short[2])      # same shape, other samples
    order = [short, long_, short2, long_, short, mid, short2, mid, long_]
    dev = torch.device('cuda', 0)
    dg = pl.DeviceGMM(g.weights_, g.means_, g.covariances_, dev)
    want = []
    for u in order:                                                          # plain passes, a context each
        p = cp.ConvertPipeline(0, FS, u, dg, order=ORDER)
        p.run()
        p.sync()
        want.append(p.wave.cpu().numpy().copy())
    pool = cp.StreamPool(0, 1)
    gen0 = pool.contexts[0].arena_generation()
    got = cp.convert_batch(order, FS, g, order=ORDER, pool=pool, shapes_per_stream=2)
    assert pool.contexts[0].arena_generation() > gen0
    for w, v in zip(want, got):
        assert np.array_equal(w, v.cpu().numpy())
    # a stale graph refuses to replay
    p = cp.ConvertPipeline(0, FS, short, dg, order=ORDER, stream=pool.streams[0], ctx=pool.contexts[0])
    p.capture()
    assert p.graph_valid()
    from kwiiyatta_amd._lib import lib, check
    check(pool.contexts[0], lib.kwy_ctx_reserve(pool.contexts[0].handle, 1 << 30))
    assert not p.graph_valid()
    with pytest.raises(RuntimeError, match='stale'):
        p.replay()


def test_training_matrix_with_device_drawn_pads(corpus):
    """pad spectra from numpy's legacy stream reproduced on the GPU (backend.nprandom) instead of np.random on the
    host: the same matrix as the host-drawn path under the same seed (the pads agree to an ulp of log(), the warping
    paths are the same), and a rank that starts in the middle of the corpus (`pairs_before`) gets the pads -- hence
    the rows -- it would get on one rank."""
    import torch
    from kwiiyatta_amd import corpus as cp
    from kwiiyatta_amd.backend.nprandom import DeviceRandomState
    np.random.seed(11)
    want, _ = cp.build_training_matrix(corpus, FS, streams=2)
    got, frames = cp.build_training_matrix(corpus, FS, streams=2, rng=DeviceRandomState.from_seed(11))
    assert frames == sum(len(s[1]) for s, _ in corpus)
    assert got.shape == want.shape
    assert torch.allclose(got, want, rtol=1e-9, atol=1e-12)
    # "rank 1 of 2": the last two pairs, the generator advanced past the first two
    tail, _ = cp.build_training_matrix(corpus[2:], FS, streams=2, rng=DeviceRandomState.from_seed(11), pairs_before=2)
    head, _ = cp.build_training_matrix(corpus[:2], FS, streams=2, rng=DeviceRandomState.from_seed(11))
    assert torch.equal(torch.cat((head, tail)), got)


def test_silence_helper_thread_stops_on_error(corpus):
    """a pair that raises must not leave the pad-drawing helper thread behind (it would keep consuming draws of the
    global generator: a retry in the same process would no longer be reproducible)"""
    import threading
    from kwiiyatta_amd import corpus as cp
    before = threading.active_count()
    bad = list(corpus[:2]) + [((None, corpus[0][0][1], corpus[0][0][2]), corpus[0][1])]   # no waveform: raises on the host
    with pytest.raises(Exception):
        cp.build_training_matrix(bad, FS, streams=1)
    assert threading.active_count() == before


@pytest.mark.parametrize('fs_u,fs_c', [(48000, 16000), (16000, 48000), (22050, 16000)])
def test_convert_pipeline_with_converter_at_another_rate(fs_u, fs_c):
    """--mcep-fs: a converter trained at another sampling rate than the utterance's.  ConvertPipeline(mcep_fs=...)
    against the package's Python API path (MelCepstrumFeatureConverter.convert, then `feature.mel_cepstrum = ...` and
    the synthesis: kwiiyatta/convert_voice.py:35-46), the silent bins of the widening step drawn from the same seeded
    numpy stream -- on the host there, on the device here."""
    import torch
    import kwiiyatta_amd as kw
    from kwiiyatta_amd import corpus as cp
    from kwiiyatta_amd import pipeline as pl
    from kwiiyatta_amd.backend import world
    from kwiiyatta_amd.backend.nprandom import DeviceRandomState
    from kwiiyatta_amd.synthetic import make_utterance
    x, _, _ = make_utterance(seed=21, fs=fs_u, seconds=0.8)
    g = pl.synthetic_gmm(order=ORDER, components=4, seed=0, n_frames=3000)
    conv = kw.MelCepstrumConverter(use_delta=True, components=4)
    conv.gmm.weights_, conv.gmm.means_, conv.gmm.covariances_ = g.weights_, g.means_, g.covariances_
    conv.order, conv.fs = ORDER, fs_c
    conv.base.frame_period = 5
    # the API path
    np.random.seed(77)
    src = kw.Analyzer(kw.Wavdata(fs_u, x), mcep_order=ORDER)
    mcep = conv.convert(src.mel_cepstrum, diff=False)
    assert mcep.fs == fs_c
    feat = kw.feature(src)
    feat.mel_cepstrum = mcep
    want = kw.Synthesizer._synthesize(feat).data
    # the device path on the same f0 track
    f0, t = np.ascontiguousarray(src.f0), np.ascontiguousarray(src._timeaxis)
    np.random.seed(77)
    dg = pl.DeviceGMM(g.weights_, g.means_, g.covariances_, torch.device('cuda', 0))
    p = cp.ConvertPipeline(0, fs_u, (x, f0, t), dg, order=ORDER, mcep_fs=fs_c, rng=DeviceRandomState.from_global())
    p.run()
    p.sync()
    got = p.wave.cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
    assert np.abs(p.mc_c2.cpu().numpy() - mcep.data).max() <= 1e-9 * np.abs(mcep.data).max()


def test_lockstep_drivers_equal_the_stream_drivers(corpus):
    """round 4's lockstep drivers (TrainWave / ConvertWave: waves of pairs or utterances through the batched entries on
    two streams, rows appended behind a device-side cursor) against round 3's stream-per-item drivers: the same
    training matrix -- several waves, host pads and device pads --, the same converted and resynthesised waveforms,
    bit for bit"""
    import torch
    from kwiiyatta_amd import corpus as cp
    from kwiiyatta_amd.backend.nprandom import DeviceRandomState
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP
    for wave_pairs in (16, 3, 1):
        np.random.seed(5)
        a, fa = cp.build_training_matrix(corpus, FS, driver='lockstep', wave_pairs=wave_pairs)
        np.random.seed(5)
        b, fb = cp.build_training_matrix(corpus, FS, driver='streams', streams=2)
        assert fa == fb and torch.equal(a, b), wave_pairs
    a, _ = cp.build_training_matrix(corpus, FS, driver='lockstep', wave_pairs=2, rng=DeviceRandomState.from_seed(3))
    b, _ = cp.build_training_matrix(corpus, FS, driver='streams', streams=2, rng=DeviceRandomState.from_seed(3))
    assert torch.equal(a, b)
    tail, _ = cp.build_training_matrix(corpus[1:], FS, driver='lockstep', rng=DeviceRandomState.from_seed(3), pairs_before=1)
    head, _ = cp.build_training_matrix(corpus[:1], FS, driver='lockstep', rng=DeviceRandomState.from_seed(3))
    assert torch.equal(torch.cat((head, tail)), a)
    g = GaussianMixtureHIP(n_components=4, max_iter=3, tol=0.0, random_state=0).fit(a)
    utts = [s for s, _ in corpus] + [t for _, t in corpus] + [corpus[0][0]] * 11      # 19: more than one wave, ragged
    w1 = cp.convert_batch(utts, FS, g, order=ORDER, driver='lockstep')
    w2 = cp.convert_batch(utts, FS, g, order=ORDER, driver='streams', streams=2)
    # (the lockstep driver takes its mel-cepstra from CheapTrick's liftered cepstrum, kwy_cheaptrick_mcep_batch_dev, the
    # stream driver from sp2mc of the envelope: equal up to the rounding of one exp / log round trip)
    for u, v in zip(w1, w2):
        assert u.shape == v.shape and float((u - v).abs().max()) <= 1e-9 * max(1.0, float(v.abs().max()))
    r1, f1 = cp.resynthesize_batch(utts, FS, driver='lockstep')
    r2, f2 = cp.resynthesize_batch(utts, FS, driver='streams', streams=2)
    assert f1 == f2
    for u, v in zip(r1, r2):
        assert torch.equal(u, v)
