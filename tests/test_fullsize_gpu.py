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
