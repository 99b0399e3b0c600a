"""Parity at the sizes bench.py actually times (BASELINE config 3): a 64-component joint GMM over D = 144
with T = 2201 frames, and one full 10 s + 11 s source/target pair through the HBM-resident PairPipeline,
every stage against the CPU oracle.  Same criteria as the miniature tests in test_pipeline_gpu.py /
test_backends_gpu.py (FastDTW path, projection and gathers bit-exact; spectra 1e-8 of the frame maximum;
aperiodicity 1e-4 absolute; MLPG 1e-9 relative; waveform 1e-9 RMS given identical features)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FS = 48000


@pytest.fixture(scope='module')
def gmm64():
    from kwiiyatta_amd import pipeline as pl
    return pl.synthetic_gmm(order=24, components=64, seed=0)


@pytest.fixture(scope='module')
def ko():
    from oracle import oracle
    return oracle


def _trajectory(T, d, seed):
    rng = np.random.default_rng(seed)
    scale = 1.0 / (1.0 + np.arange(d)) ** 0.7
    walk = np.cumsum(rng.standard_normal((T, d)), axis=0) * 0.05
    return np.ascontiguousarray((walk - walk.mean(0)) * scale + rng.standard_normal((T, d)) * 0.02 * scale)


@pytest.mark.parametrize('diff', [False, True])
def test_gmm_mlpg_bench_size(ko, gmm64, diff):
    """kwy_gmm_mlpg with the grid bench.py launches: M = 64, D = 144, T = 2201."""
    from kwiiyatta_amd.backend import mlpg
    mc = _trajectory(2201, 24, seed=11)
    X = mlpg.delta_features(mc, mlpg.DELTA_WINDOWS)
    ref, mix = ko.gmm_mlpg(mc, gmm64.weights_, gmm64.means_, gmm64.covariances_, diff=diff, return_mix=True)
    got = mlpg.MLPG(gmm64, windows=mlpg.DELTA_WINDOWS, diff=diff).transform(X)
    assert got.shape == ref.shape == (2201, 24)
    assert np.abs(got - ref).max() <= 1e-10 * max(np.abs(ref).max(), 1.0)
    assert len(np.unique(mix)) >= 4             # the arg-max selection really switches mixtures


def test_pair_pipeline_full_size(ko, gmm64):
    """10 s source (T = 2001) + 11 s target (T = 2201), M = 64: what one bench step does for one pair."""
    import torch
    from kwiiyatta_amd import pipeline as pl
    from kwiiyatta_amd.synthetic import make_utterance
    from kwiiyatta_amd.vocoder.align import project_path_iter
    src = make_utterance(seed=1234, fs=FS, seconds=10.0)
    tgt = make_utterance(seed=4321, fs=FS, seconds=10.0, time_warp=1.1, formant_scale=1.12)
    assert len(src[1]) == 2001 and len(tgt[1]) == 2201
    dg = pl.DeviceGMM(gmm64.weights_, gmm64.means_, gmm64.covariances_, torch.device('cuda', 0))
    p = pl.PairPipeline(0, FS, src, tgt, dg)
    p.run()
    p.sync()
    P = pl.PAD_LEN
    alpha = ko.mcepalpha(FS)
    for side, (x, f0, t) in ((p.src, src), (p.tgt, tgt)):
        sp_pad = side.sp_pad.cpu().numpy()
        ap_pad = side.ap_pad.cpu().numpy()
        sp_ref = ko.cheaptrick(x, f0, t, FS) / FS
        d = np.abs(sp_pad[P:P + len(f0)] - sp_ref)
        assert d.max() <= 1e-8 * sp_ref.max() and d.sum() <= 1e-9 * sp_ref.sum()
        assert np.abs(ap_pad[P:P + len(f0)] - ko.d4c(x, f0, t, FS)).max() <= 1e-4
        mc_ref = ko.sp2mc(sp_pad, 24, alpha)
        mc = side.mc_pad.cpu().numpy()
        assert np.abs(mc - mc_ref).max() <= 1e-11 * np.abs(mc_ref).max()
    feat_s, feat_t = p.src.feat.cpu().numpy(), p.tgt.feat.cpu().numpy()
    d_ref, path_ref = ko.fastdtw(feat_s, feat_t, radius=32, dist=2)
    n = int(p.path_len.item())
    assert [tuple(r) for r in p.path.cpu().numpy()[:n].tolist()] == path_ref and p.dist.item() == d_ref
    idx_ref = list(project_path_iter(np.array(path_ref), trim=True, trim_len=P))
    idx = p.idx.cpu().numpy()[:p.tgt.T]
    assert int(p.n_idx.item()) == len(idx_ref) == p.tgt.T and idx.tolist() == idx_ref
    mc_al = p.mc_al.cpu().numpy()
    assert np.array_equal(mc_al, p.src.mc_pad.cpu().numpy()[idx])
    assert np.array_equal(p.ap_al.cpu().numpy(), p.src.ap_pad.cpu().numpy()[idx])
    y_ref, mix = ko.gmm_mlpg(np.ascontiguousarray(mc_al[:, 1:]), gmm64.weights_, gmm64.means_,
                             gmm64.covariances_, return_mix=True)
    assert len(np.unique(mix)) >= 4
    mc_conv = p.mc_conv.cpu().numpy()
    assert np.array_equal(mc_conv[:, 0], mc_al[:, 0])
    assert np.abs(mc_conv[:, 1:] - y_ref).max() <= 1e-9 * max(np.abs(y_ref).max(), 1)
    sp_conv = p.sp_conv.cpu().numpy()
    assert np.max(np.abs(sp_conv / ko.mc2sp(mc_conv, alpha, 2048) - 1)) <= 1e-10
    wave_ref = ko.synthesize(tgt[1], np.ascontiguousarray(sp_conv * FS), p.ap_al.cpu().numpy(), FS, 5.0)
    wave = p.wave.cpu().numpy()
    assert len(wave) == len(wave_ref) == 528240
    assert np.sqrt(np.mean((wave - wave_ref) ** 2)) <= 1e-9


def test_pair_chained_against_all_oracle_chain(ko, gmm64):
    """North star, config 3: the pipeline's FINAL waveform against an all-CPU chain in which every oracle stage is
    fed by the oracle's own previous output (not by the GPU's), on the same waveforms, f0 tracks, GMM and the same
    four pad blocks.  Equal FastDTW path, equal projection, waveform within 1e-4 RMS."""
    import torch
    from oracle import chain
    from kwiiyatta_amd import pipeline as pl
    from kwiiyatta_amd.synthetic import make_utterance
    src = make_utterance(seed=1234, fs=FS, seconds=10.0)
    tgt = make_utterance(seed=4321, fs=FS, seconds=10.0, time_warp=1.1, formant_scale=1.12)
    rng = np.random.RandomState(7)
    silence = [chain.draw_silence(rng, FS, 1025) for _ in range(4)]
    ref = chain.pair_chain(src, tgt, (gmm64.weights_, gmm64.means_, gmm64.covariances_), FS, silence)
    dg = pl.DeviceGMM(gmm64.weights_, gmm64.means_, gmm64.covariances_, torch.device('cuda', 0))
    p = pl.PairPipeline(0, FS, src, tgt, dg, silence=silence)
    p.run()
    p.sync()
    n = int(p.path_len.item())
    path = [tuple(r) for r in p.path.cpu().numpy()[:n].tolist()]
    differing = [(a, b) for a, b in zip(path, ref['path']) if a != b]
    assert len(path) == len(ref['path']) and not differing, f'{len(differing)} path cells differ: {differing[:5]}'
    assert p.dist.item() == pytest.approx(ref['dist'], rel=1e-9)
    assert p.idx.cpu().numpy()[:p.tgt.T].tolist() == ref['idx'].tolist()
    ap_err = np.abs(p.ap_al.cpu().numpy() - ref['ap_al']).max()
    mc_err = np.abs(p.mc_conv.cpu().numpy() - ref['mc_conv']).max()
    wave = p.wave.cpu().numpy()
    assert len(wave) == len(ref['wave'])
    rms = float(np.sqrt(np.mean((wave - ref['wave']) ** 2)))
    peak = float(np.abs(ref['wave']).max())
    print(f'chained config 3: wave rms {rms:.3e} (peak {peak:.3f}), aligned ap max err {ap_err:.3e}, '
          f'converted mcep max err {mc_err:.3e}')
    assert ap_err <= 1e-4 and mc_err <= 1e-8
    assert rms <= 1e-4


@pytest.mark.parametrize('which', ['16k', '48k'])
def test_pair_chained_on_recorded_speech(ko, which):
    """The same chained comparison on RECORDED speech (two CMU ARCTIC speakers saying the same sentence; the 48 kHz
    variant of the reference's own fixtures for the second case): f0 by the library's DIO + StoneMask, an 8-component
    GMM, the pipeline's final waveform against the all-oracle chain.  Real recordings have frames near the voicing
    gates (D4C's LoveTrain threshold, aperiodicity 0.999) that the synthetic signals lack."""
    import torch
    from scipy.io import wavfile
    from conftest import CLB_WAV, SLT_WAV, clb_variant
    from oracle import chain
    from kwiiyatta_amd import pipeline as pl
    from kwiiyatta_amd.backend import world

    def load(path):
        fs, d = wavfile.read(path)
        return fs, np.ascontiguousarray(d.astype(np.float64) / 2 ** 15)
    if which == '16k':
        (fs, xs), (_, xt) = load(CLB_WAV), load(SLT_WAV)
    else:
        fs, xs = load(clb_variant('48'))
        xt = np.ascontiguousarray(xs[int(0.05 * fs):] * 0.8)      # the same recording, shifted and scaled: a second "speaker"
    utts = []
    for x in (xs, xt):
        f0, t = world.dio(x, fs, frame_period=5.0)
        utts.append((x, world.stonemask(x, f0, t, fs), t))
    g = pl.synthetic_gmm(order=24, components=8, seed=0, n_frames=4000)
    K = ko.get_cheaptrick_fft_size(fs) // 2 + 1
    rng = np.random.RandomState(3)
    silence = [chain.draw_silence(rng, fs, K) for _ in range(4)]
    ref = chain.pair_chain(utts[0], utts[1], (g.weights_, g.means_, g.covariances_), fs, silence)
    dg = pl.DeviceGMM(g.weights_, g.means_, g.covariances_, torch.device('cuda', 0))
    p = pl.PairPipeline(0, fs, utts[0], utts[1], dg, silence=silence)
    p.run()
    p.sync()
    n = int(p.path_len.item())
    path = [tuple(r) for r in p.path.cpu().numpy()[:n].tolist()]
    assert path == ref['path'], f'{sum(a != b for a, b in zip(path, ref["path"]))} of {len(ref["path"])} path cells differ'
    wave = p.wave.cpu().numpy()
    rms = float(np.sqrt(np.mean((wave - ref['wave']) ** 2)))
    ap_err = float(np.abs(p.ap_al.cpu().numpy() - ref['ap_al']).max())
    print(f'chained, recorded speech at {fs} Hz: wave rms {rms:.3e} (peak {np.abs(ref["wave"]).max():.3f}), '
          f'aligned aperiodicity max err {ap_err:.3e}')
    assert rms <= 1e-4


# ---------------------------------------------------------------------------------------------------------------
# Round 4: the sizes BASELINE.json's configs name, driver-run (not only builder-run lines under profiles/)

def _make_utterance_job(job):
    """(seed, seconds, f0_base, warp, formant) -> utterance.  Module-level for the process pool; synthetic.py is loaded
    by path, so the (spawned) workers need neither the package nor the HIP runtime."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('kwy_synthetic', os.path.join(root, 'kwiiyatta_amd', 'synthetic.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seed, seconds, f0_base, warp, formant = job
    return mod.make_utterance(seed=seed, fs=FS, seconds=seconds, f0_base=f0_base, time_warp=warp, formant_scale=formant)


def _generate(jobs):
    import concurrent.futures as cf
    import multiprocessing as mp
    import os
    nproc = max(1, min(len(os.sched_getaffinity(0)), 16, len(jobs)))
    with cf.ProcessPoolExecutor(nproc, mp_context=mp.get_context('spawn')) as ex:
        return list(ex.map(_make_utterance_job, jobs, chunksize=2))


def test_lockstep_step_full_size_against_all_oracle_chain(ko, gmm64):
    """config 3 on the lockstep driver at BASELINE size: four 10 s + 11 s pairs in two waves, the pads drawn inside the
    step by the device generator.  Pair 0 against the all-oracle chain ON THE PADS THE DEVICE DREW (equal FastDTW path,
    waveform within 1e-4 RMS), the generator equal to numpy's afterwards, a replayed graph bit-identical in everything
    but the fresh pads' effect (checked: same path, waveform within 1e-9 of the first pass)."""
    import torch
    from oracle import chain
    from kwiiyatta_amd import pipeline as pl
    from kwiiyatta_amd.backend.nprandom import DeviceRandomState
    utts = _generate([(1234 + 2 * i, 10.0, 140.0, 1.0, 1.0) for i in range(4)] +
                     [(4321 + 2 * i, 10.0, 140.0, 1.1, 1.12) for i in range(4)])
    pairs = [(utts[i], utts[4 + i]) for i in range(4)]
    assert len(pairs[0][0][1]) == 2001 and len(pairs[0][1][1]) == 2201
    dg = pl.DeviceGMM(gmm64.weights_, gmm64.means_, gmm64.covariances_, torch.device('cuda', 0))
    rs = DeviceRandomState.from_seed(99)
    b = pl.PairBatchPipeline(0, FS, pairs, dg, waves=2, rng=rs)
    b.run()
    b.sync()
    pads0 = [blk.cpu().numpy() for blk in b.pad_rows[:4]]
    ref_rng = np.random.RandomState(99)
    for blk in b.pad_rows:
        exp = np.abs(ref_rng.normal(0, pl.EPS / FS, (pl.PAD_LEN, 1025)))
        assert np.abs(blk.cpu().numpy() - exp).max() <= 4e-16 * exp.max()
    st_d, st_n = rs.get_state(), ref_rng.get_state()
    assert np.array_equal(st_d[1], st_n[1]) and st_d[2:] == tuple(st_n[2:])
    ref = chain.pair_chain(pairs[0][0], pairs[0][1], (gmm64.weights_, gmm64.means_, gmm64.covariances_), FS, pads0)
    path_t, n_t, _ = b.path(0)
    path = [tuple(r) for r in path_t.cpu().numpy()[:int(n_t.item())].tolist()]
    assert path == ref['path']
    wave = b.wave(0).cpu().numpy()
    rms = float(np.sqrt(np.mean((wave - ref['wave']) ** 2)))
    print(f'lockstep config 3, pair 0 vs the all-oracle chain on device-drawn pads: wave rms {rms:.3e}')
    assert rms <= 1e-4
    first = [b.wave(k).clone() for k in range(4)]
    b.capture()
    b.replay()
    b.sync()
    for k in range(4):
        assert torch.isfinite(b.wave(k)).all()
        assert float((b.wave(k) - first[k]).abs().max()) <= 1e-9      # other pads (~1e-21 spectra): same alignment


def test_config4_256_utterances_of_10_seconds(ko):
    """BASELINE config 4 at its own size: 256 distinct 48 kHz utterances of 10 s (T = 2001 each) analysed and
    resynthesised in lockstep waves of 16 (corpus.resynthesize_batch).  All outputs finite, a second pass
    bit-identical, four utterances against the oracle's own analyse -> synthesise chain (north star: 1e-4 RMS)."""
    import torch
    from kwiiyatta_amd import corpus as cp
    n_utt = 256
    utts = _generate([(s, 10.0, 110.0 + (s % 7) * 15.0, 1.0, 1.0) for s in range(n_utt)])
    dev = torch.device('cuda', 0)
    resident = [tuple(torch.from_numpy(a).to(dev) for a in u) for u in utts]     # inputs live in HBM
    ls = cp._Lockstep(0)
    waves, frames = cp.resynthesize_batch(resident, FS, lockstep=ls)
    assert frames == n_utt * 2001
    picks = (0, 17, 100, 255)
    first = {i: waves[i].cpu().numpy().copy() for i in picks}
    finite = all(bool(torch.isfinite(w).all().item()) and float(w.abs().max().item()) > 1e-3 for w in waves)
    assert finite
    sums = [float(w.sum().item()) for w in waves]
    waves2, _ = cp.resynthesize_batch(resident, FS, lockstep=ls)
    assert all(torch.equal(a, b) for a, b in zip(waves, waves2)) and sums == [float(w.sum().item()) for w in waves2]
    worst = 0.0
    for i in picks:
        x, f0, t = utts[i]
        sp = ko.cheaptrick(x, f0, t, FS)
        ap = ko.d4c(x, f0, t, FS)
        ref = ko.synthesize(f0, sp, ap, FS, 5.0)
        assert ref.shape == first[i].shape
        worst = max(worst, float(np.sqrt(np.mean((first[i] - ref) ** 2))))
    print(f'config 4 at size: 256 x 10 s in lockstep waves, worst RMS against the all-oracle chain {worst:.3e}')
    assert worst <= 1e-4


def _corpus_rank(rank, world, port, q, jobs):
    """one rank of the config-5 data-set phase: its contiguous block of the corpus, the ONE generator stream of the
    corpus advanced past the blocks before it"""
    import hashlib
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from kwiiyatta_amd import corpus as cp
    from kwiiyatta_amd.backend.nprandom import DeviceRandomState
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    utts = [_make_utterance_job(j) for j in jobs]
    n = len(utts) // 2
    corpus = [(utts[i], utts[n + i]) for i in range(n)]
    mine = cp.shard_block(n, rank, world)
    X, frames = cp.build_training_matrix([corpus[i] for i in mine], FS, rng=DeviceRandomState.from_seed(2024),
                                         pairs_before=mine[0])
    h = hashlib.sha256(X.cpu().numpy().tobytes()).hexdigest()
    q.put((rank, h, int(X.shape[0]), float(X.sum().item()), frames))
    dist.barrier()
    dist.destroy_process_group()


def test_config5_64_distinct_pairs_one_and_two_ranks(gmm64):
    """BASELINE config 5 at 64 DISTINCT 48 kHz pairs (2 s each): the training matrix of two ranks (gloo, both on
    cuda:0, contiguous blocks, each advancing the corpus' one generator stream past the block before it) is the
    one-rank matrix, byte for byte; the fit runs real EM iterations on it; the batch conversion is finite and
    deterministic."""
    import hashlib
    import multiprocessing as mp
    import socket
    import torch
    from kwiiyatta_amd import corpus as cp
    from kwiiyatta_amd.backend.nprandom import DeviceRandomState
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP
    n = 64
    jobs = [(1000 + k, 2.0, 100.0 + (k * 37 % 90), 1.0, 1.0) for k in range(n)] + \
           [(5000 + k, 2.0, (100.0 + (k * 37 % 90)) * 1.25, 1.04 + 0.01 * (k % 9), 1.08 + 0.01 * (k % 7)) for k in range(n)]
    utts = _generate(jobs)
    corpus = [(utts[i], utts[n + i]) for i in range(n)]
    X, frames = cp.build_training_matrix(corpus, FS, rng=DeviceRandomState.from_seed(2024))
    assert frames == sum(len(s[1]) for s, _ in corpus) and X.shape[1] == 144 and X.shape[0] > 0.5 * frames
    whole = X.cpu().numpy()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    procs = [ctx.Process(target=_corpus_rank, args=(r, 2, port, q, jobs)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows0, rows1 = res[0][2], res[1][2]
    assert rows0 + rows1 == whole.shape[0] and res[0][4] + res[1][4] == frames
    assert res[0][1] == hashlib.sha256(whole[:rows0].tobytes()).hexdigest()
    assert res[1][1] == hashlib.sha256(whole[rows0:].tobytes()).hexdigest()
    g = GaussianMixtureHIP(n_components=16, max_iter=30, tol=1e-3, random_state=0).fit(X)
    assert g.n_iter_ >= 3 and np.isfinite(g.lower_bound_)
    sources = [s for s, _ in corpus]
    w1 = cp.convert_batch(sources, FS, g)
    w2 = cp.convert_batch(sources, FS, g)
    assert all(bool(torch.isfinite(a).all().item()) and torch.equal(a, b) for a, b in zip(w1, w2))


def test_config5_at_the_size_the_baseline_names():
    """BASELINE config 5 at ITS size: 503 distinct 48 kHz pairs of 5 s (an atr503-sized parallel corpus), M = 64, the
    reference's stopping rule (GaussianMixture(max_iter=100, tol=1e-3), /root/reference/kwiiyatta/converter/gmm.py:14-26).
    Too large for an oracle run, so properties: the training matrix of two independent builds is the same bytes
    (SHA-256) with the same row count; the fit converges under the rule; the batch conversion of 64 sources is finite,
    audible and bit-identical on a second pass.  (~25 s of host signal generation in a process pool first.)"""
    import hashlib
    import torch
    from kwiiyatta_amd import corpus as cp
    from kwiiyatta_amd.backend.nprandom import DeviceRandomState
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP
    n = 503
    jobs = [(1000 + k, 5.0, 100.0 + (k * 37 % 90), 1.0, 1.0) for k in range(n)] + \
           [(5000 + k, 5.0, (100.0 + (k * 37 % 90)) * (1.25 + 0.01 * (k % 11)), 1.04 + 0.01 * (k % 9), 1.08 + 0.01 * (k % 7))
            for k in range(n)]
    utts = _generate(jobs)
    corpus = [(utts[i], utts[n + i]) for i in range(n)]
    ls = cp._Lockstep(0)
    X, frames = cp.build_training_matrix(corpus, FS, rng=DeviceRandomState.from_seed(1234), driver='lockstep', lockstep=ls)
    assert frames == sum(len(s[1]) for s, _ in corpus) == n * 1001
    assert X.shape[1] == 144 and 0.5 * frames < X.shape[0] < 2 * frames
    h1 = hashlib.sha256(X.cpu().numpy().tobytes()).hexdigest()
    X2, _ = cp.build_training_matrix(corpus, FS, rng=DeviceRandomState.from_seed(1234), driver='lockstep', lockstep=ls)
    assert X2.shape == X.shape and hashlib.sha256(X2.cpu().numpy().tobytes()).hexdigest() == h1
    del X2
    assert bool(torch.isfinite(X).all().item())
    g = GaussianMixtureHIP(n_components=64, max_iter=100, tol=1e-3, random_state=0).fit(X)
    print(f'config 5 at size: {X.shape[0]} rows, {g.kmeans_n_iter_} Lloyd + {g.n_iter_} EM iterations, '
          f'lower bound {g.lower_bound_:.6f}, converged {g.converged_}')
    assert g.converged_ and 2 <= g.n_iter_ <= 100 and np.isfinite(g.lower_bound_)
    assert np.all(np.isfinite(g.means_)) and abs(float(np.sum(g.weights_)) - 1.0) < 1e-9
    del X
    sources = [s for s, _ in corpus[:64]]
    w1 = cp.convert_batch(sources, FS, g, lockstep=ls)
    sums = [float(w.abs().max().item()) for w in w1]
    keep = [w.clone() for w in w1]
    w2 = cp.convert_batch(sources, FS, g, lockstep=ls)
    assert all(bool(torch.isfinite(a).all().item()) and torch.equal(a, b) for a, b in zip(keep, w2))
    assert min(sums) > 1e-3
