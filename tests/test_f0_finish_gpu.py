"""Round 5: the two ends of the path on the device.

* kwy_dio_batch_dev / kwy_stonemask_batch_dev (device pointers, <= 16 utterances per pass of launches): every
  utterance's track equals the host entry's bit for bit (the host entries run the same pass on staged copies), and the
  host entries are checked against the oracle in test_backends_gpu.py and here on a 48 kHz signal.  Reference:
  /root/reference/kwiiyatta/vocoder/world.py:33-41.
* kwy_finish_pcm16_batch_dev: int16 samples equal to Synthesizer.finish + Wavdata.save's on the host
  (/root/reference/kwiiyatta/vocoder/abc/synthesizer.py:11-20, wavfile.py:8-29), which needs numpy's chunked pairwise
  mean to the last bit."""
import numpy as np
import pytest

from conftest import CLB_WAV, CLB_WAV2, SLT_WAV, clb_variant

pytestmark = pytest.mark.gpu


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _load(path):
    from kwiiyatta_amd.wavfile import load_wav
    w = load_wav(path)
    return w.fs, np.ascontiguousarray(w.data)


def _batch_f0(ctx, waves, fs, frame_period=5.0):
    import torch
    from kwiiyatta_amd.backend import world
    T = [world.dio_frames(fs, len(x), frame_period) for x in waves]
    dx = [_dev(x) for x in waves]
    t = [torch.empty(n, dtype=torch.float64, device='cuda') for n in T]
    f0 = [torch.empty(n, dtype=torch.float64, device='cuda') for n in T]
    refined = [torch.empty(n, dtype=torch.float64, device='cuda') for n in T]
    status = torch.full((len(waves),), 7, dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()          # (torch filled these on ITS stream; the library runs on the context's)
    world.dio_batch_dev(ctx, dx, fs, t, f0, status, frame_period=frame_period)
    world.stonemask_batch_dev(ctx, dx, t, f0, fs, refined)
    ctx.sync()
    return ([a.cpu().numpy() for a in t], [a.cpu().numpy() for a in f0], [a.cpu().numpy() for a in refined],
            status.cpu().numpy())


@pytest.mark.parametrize('frame_period', [5.0, 3.0])
def test_f0_batch_equals_host_entries_16k(frame_period):
    """19 ragged utterances (two passes of launches): recorded speech, cuts of it, silence, a signal shorter than the
    voiced-range minimum"""
    from kwiiyatta_amd import _lib
    from kwiiyatta_amd.backend import world
    fs, a = _load(CLB_WAV)
    _, b = _load(CLB_WAV2)
    _, c = _load(SLT_WAV)
    rng = np.random.default_rng(5)
    waves = [a, b, c, a[:20000], b[3000:41000], c[:8191], a[:8192], b[:8193], np.zeros(3000), rng.standard_normal(800) * 0.1,
             c[10000:], a[::-1].copy(), b[:30001], c[:12345], a[5000:25000], b[:16384], c[:50000], a[:777], b[100:9000]]
    ctx = _lib.Context(0)
    t, f0, refined, status = _batch_f0(ctx, waves, fs, frame_period)
    assert not status.any()
    for i, x in enumerate(waves):
        f0_h, t_h = world.dio(x, fs, frame_period=frame_period)
        assert np.array_equal(t[i], t_h), i
        assert np.array_equal(f0[i], f0_h), i
        assert np.array_equal(refined[i], world.stonemask(x, f0_h, t_h, fs)), i


@pytest.mark.parametrize('suffix', ['22', '48', '96'])
def test_f0_batch_equals_host_entries_other_rates(suffix):
    from kwiiyatta_amd import _lib
    from kwiiyatta_amd.backend import world
    fs, a = _load(clb_variant(suffix))
    waves = [a, a[:len(a) // 2], a[len(a) // 3:], a[:fs]]
    ctx = _lib.Context(0)
    t, f0, refined, status = _batch_f0(ctx, waves, fs)
    assert not status.any()
    for i, x in enumerate(waves):
        f0_h, t_h = world.dio(x, fs)
        assert np.array_equal(t[i], t_h) and np.array_equal(f0[i], f0_h), i
        assert np.array_equal(refined[i], world.stonemask(x, f0_h, t_h, fs)), i


def test_f0_batch_synthetic_48k_vs_oracle():
    """the benchmark's signal family: DIO + StoneMask of a batch against the CPU oracle (same voicing, 1e-8 Hz)"""
    from oracle import oracle as ko
    from kwiiyatta_amd import _lib
    from kwiiyatta_amd.synthetic import make_utterance
    fs = 48000
    waves = [make_utterance(seed=s, fs=fs, seconds=sec)[0] for s, sec in ((3, 1.6), (4, 2.0), (5, 1.0))]
    ctx = _lib.Context(0)
    t, f0, refined, status = _batch_f0(ctx, waves, fs)
    assert not status.any()
    for i, x in enumerate(waves):
        f0_ref, t_ref = ko.dio(x, fs)
        assert np.array_equal(t[i], t_ref)
        assert np.array_equal(f0[i] > 0, f0_ref > 0)
        assert np.abs(f0[i] - f0_ref).max() <= 1e-8
        s_ref = ko.stonemask(x, f0[i], t_ref, fs)          # the oracle's refinement of the device's own DIO track
        assert np.array_equal(refined[i] > 0, s_ref > 0)
        assert np.abs(refined[i] - s_ref).max() <= 1e-10 * s_ref.max()


def test_dio_dev_single_and_status_word():
    import torch
    from kwiiyatta_amd import _lib
    from kwiiyatta_amd._lib import lib, c_vp
    from kwiiyatta_amd.backend import world
    fs, a = _load(CLB_WAV)
    ctx = _lib.Context(0)
    T = world.dio_frames(fs, len(a))
    dx = _dev(a)
    t = torch.empty(T, dtype=torch.float64, device='cuda')
    f0 = torch.empty(T, dtype=torch.float64, device='cuda')
    st = torch.full((1,), 5, dtype=torch.int32, device='cuda')
    torch.cuda.synchronize()
    _lib.check(ctx, lib.kwy_dio_dev(ctx.handle, c_vp(dx.data_ptr()), len(a), fs, 71.0, 800.0, 2.0, 5.0, 1, 0.1,
                                    c_vp(t.data_ptr()), c_vp(f0.data_ptr()), c_vp(st.data_ptr())))
    out = torch.empty(T, dtype=torch.float64, device='cuda')
    _lib.check(ctx, lib.kwy_stonemask_dev(ctx.handle, c_vp(dx.data_ptr()), len(a), fs, c_vp(t.data_ptr()),
                                          c_vp(f0.data_ptr()), T, c_vp(out.data_ptr())))
    ctx.sync()
    f0_h, t_h = world.dio(a, fs)
    assert int(st.item()) == 0
    assert np.array_equal(f0.cpu().numpy(), f0_h) and np.array_equal(t.cpu().numpy(), t_h)
    assert np.array_equal(out.cpu().numpy(), world.stonemask(a, f0_h, t_h, fs))
    with pytest.raises(ValueError):
        _lib.check(ctx, lib.kwy_dio_dev(ctx.handle, c_vp(dx.data_ptr()), len(a), fs, 71.0, 800.0, 2.0, 5.0, 2, 0.1,
                                        c_vp(t.data_ptr()), c_vp(f0.data_ptr()), None))


# ---------------------------------------------------------------------------------------------------- post-step
def _host_pcm(y, frame_len, fs, normalize_synth=True, synth_kw=None, normalize_save=True, save_kw=None):
    """the package's host path: Synthesizer.finish (the reference's loop) + Wavdata._pcm16"""
    import kwiiyatta_amd as k
    w = k.Wavdata(fs, y.copy())
    if normalize_synth:
        k.Synthesizer.finish(w, frame_len, **(synth_kw or {}))
    return w._pcm16(normalize_save, save_kw or {})


def _device_pcm(ctx, ys, frame_lens, fs, **kw):
    import torch
    from kwiiyatta_amd.backend import finish
    dy = [_dev(y) for y in ys]
    keep = [y.clone() for y in dy]
    pcm = [torch.full((len(y),), -7, dtype=torch.int16, device='cuda') for y in ys]
    torch.cuda.synchronize()
    finish.pcm16_batch_dev(ctx, dy, frame_lens, fs, pcm, **kw)
    ctx.sync()
    for a, b in zip(dy, keep):
        assert torch.equal(a, b)                 # the waveform itself is not modified
    return [p.cpu().numpy() for p in pcm]


@pytest.mark.parametrize('fs', [48000, 44100, 16000])
def test_finish_pcm16_equals_host_path(fs):
    from kwiiyatta_amd import _lib
    rng = np.random.default_rng(fs)
    lens = [5, 8, 127, 128, 129, 1000, 8191, 8192, 8193, 16384 + 77, 100003, 131072, 3 * 8192 + 135, 240240, 52801,
            8192 * 5 + 4103, 99, 31000, 65536 + 8]
    ys, frames = [], []
    for i, n in enumerate(lens):
        y = rng.standard_normal(n) * rng.uniform(0.05, 0.4) + rng.uniform(-0.2, 0.2)
        # loud stretches: some pieces above the ceiling, and (every third) a global peak above it
        for _ in range(3):
            a = int(rng.integers(0, n))
            y[a:a + max(1, n // 50)] *= 6.0 if i % 3 == 0 else 2.5
        ys.append(y)
        ms = n * 1000 // fs                          # whole milliseconds of signal
        frames.append(int(min(ms, max(0, n // (fs // 200)))))     # at most one piece per millisecond of signal
    ctx = _lib.Context(0)
    got = _device_pcm(ctx, ys, frames, fs)
    for i, (y, T) in enumerate(zip(ys, frames)):
        exp = _host_pcm(y, T, fs)
        assert exp.dtype == np.int16
        assert np.array_equal(got[i], exp), (i, lens[i], int(np.abs(got[i].astype(int) - exp.astype(int)).max()))


def test_finish_pcm16_options_and_errors():
    from kwiiyatta_amd import _lib
    fs = 22050
    rng = np.random.default_rng(9)
    ys = [rng.standard_normal(n) * 0.5 + 0.1 for n in (30000, 12345, 9000)]
    frames = [200, 90, 300]
    ctx = _lib.Context(0)
    for kw_dev, kw_host in (
            (dict(save_peak_lv=None), dict(save_kw=dict(peak_lv=None))),
            (dict(normalize_save=False), dict(normalize_save=False)),
            (dict(normalize_synth=False), dict(normalize_synth=False)),
            (dict(synth_peak_lv=-6, save_peak_lv=-3), dict(synth_kw=dict(peak_lv=-6), save_kw=dict(peak_lv=-3))),
            (dict(normalize_synth=False, normalize_save=False), dict(normalize_synth=False, normalize_save=False))):
        got = _device_pcm(ctx, ys, frames, fs, **kw_dev)
        for y, T, g in zip(ys, frames, got):
            assert np.array_equal(g, _host_pcm(y, T, fs, **kw_host)), (kw_dev,)
    # more frames than milliseconds of signal: the reference's loop fails on the empty piece -- so does the entry
    with pytest.raises(ValueError):
        _device_pcm(ctx, [ys[2]], [9000 * 1000 // fs + 2], fs)
    import kwiiyatta_amd as k
    with pytest.raises(ValueError):
        k.Synthesizer.finish(k.Wavdata(fs, ys[2].copy()), 9000 * 1000 // fs + 2)


def test_finish_on_synthesised_speech():
    """the real thing: analyse + synthesise a recording on the GPU, post-step on the device == on the host"""
    import kwiiyatta_amd as k
    from kwiiyatta_amd import _lib
    a = k.analyze_wav(CLB_WAV)
    feat = k.feature(a)
    wav = feat.Synthesizer.synthesize(feat, normalize=False)
    ctx = _lib.Context(0)
    got = _device_pcm(ctx, [np.ascontiguousarray(wav.data)], [feat.frame_len], wav.fs)[0]
    exp = _host_pcm(np.ascontiguousarray(wav.data), feat.frame_len, wav.fs)
    assert np.array_equal(got, exp)
    assert np.abs(exp).max() > 1000


def test_convert_batch_wav_in_pcm_out():
    """corpus.convert_batch on bare waveforms with pcm=True (what `kwiiyatta --batch` runs): the waveforms equal the
    (x, f0, t) form's bit for bit -- the f0 track extracted inside the wave is the Analyzer's --, and the int16 samples
    equal Synthesizer.finish + Wavdata.save of those waveforms on the host.  19 files' worth: two waves."""
    import kwiiyatta_amd as k
    from kwiiyatta_amd import corpus, pipeline as pl
    paths = [CLB_WAV, CLB_WAV2, SLT_WAV]
    an = [k.analyze_wav(p) for p in paths]
    fs = an[0].fs
    waves_in = [np.ascontiguousarray(a.wavdata.data) for a in an]
    waves_in = (waves_in * 7)[:19]
    waves_in = [w[:len(w) - 137 * i] for i, w in enumerate(waves_in)]          # ragged
    gmm = pl.synthetic_gmm(order=24, components=4, seed=1, n_frames=4000)
    got, pcm = corpus.convert_batch(waves_in, fs, gmm, pcm=True)
    triples = []
    from kwiiyatta_amd.backend import world
    for w in waves_in:
        f0c, t = world.dio(w, fs)
        triples.append((w, world.stonemask(w, f0c, t, fs), t))
    ref = corpus.convert_batch(triples, fs, gmm)
    for i, (a, b, p) in enumerate(zip(got, ref, pcm)):
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert np.array_equal(a, b), i
        assert np.isfinite(a).all() and np.abs(a).max() > 1e-3
        assert np.array_equal(p.cpu().numpy(), _host_pcm(a, len(triples[i][1]), fs)), i


def test_mlsa_filter_batch_equals_single_calls():
    """kwy_mlsa_filter_batch_dev (one wavefront per signal, all signals of a launch side by side; more jobs than one
    launch holds) == kwy_mc2b_dev + kwy_mlsa_synthesis_dev per signal, bit for bit; with and without c0"""
    import torch
    from kwiiyatta_amd import _lib
    from kwiiyatta_amd._lib import lib, c_vp
    rng = np.random.default_rng(3)
    order, alpha, hop, pd = 24, 0.55, 240, 4
    ctx = _lib.Context(0)
    jobs, single = [], []
    keep = []
    for i in range(70):
        T = int(rng.integers(3, 60))
        n = T * hop + int(rng.integers(-hop + 1, hop))
        x = _dev(rng.standard_normal(n) * 0.3)
        mc = _dev(rng.standard_normal((T, order + 1)) * 0.05)
        y = torch.full((n,), 9.0, dtype=torch.float64, device='cuda')
        keep.append((x, mc, y))
        jobs.append((x, n, mc, T, y))
    torch.cuda.synchronize()
    for ignore in (1, 0):
        arr = _lib.job_array(_lib.MlsaJob, jobs)
        _lib.check(ctx, lib.kwy_mlsa_filter_batch_dev(ctx.handle, arr, len(jobs), order, alpha, pd, hop, ignore))
        ctx.sync()
        for x, n, mc, T, y in jobs[::7]:
            m2 = mc.clone()
            if ignore:
                m2[:, 0] = 0.0
            b = torch.empty_like(m2)
            y1 = torch.empty_like(y)
            torch.cuda.synchronize()          # (torch prepared m2 on ITS stream; the library runs on the context's)
            _lib.check(ctx, lib.kwy_mc2b_dev(ctx.handle, c_vp(m2.data_ptr()), T, order, alpha, c_vp(b.data_ptr())))
            _lib.check(ctx, lib.kwy_mlsa_synthesis_dev(ctx.handle, c_vp(x.data_ptr()), n, c_vp(b.data_ptr()), T, order, alpha,
                                                       pd, hop, c_vp(y1.data_ptr())))
            ctx.sync()
            assert torch.equal(y, y1)
            assert bool(torch.isfinite(y).all())


def test_convert_batch_diff_outputs():
    """corpus.convert_batch(diff=True): the differential output of every file equals apply_mlsa_filter of the
    differential conversion through the Python API's stages (same kernels, single calls) bit for bit, and its int16
    samples equal Wavdata.save's"""
    import torch
    import kwiiyatta_amd as k
    from kwiiyatta_amd import _lib, corpus, pipeline as pl
    from kwiiyatta_amd._lib import lib, c_vp
    an = [k.analyze_wav(p) for p in (CLB_WAV, SLT_WAV, CLB_WAV2)]
    fs = an[0].fs
    waves_in = [np.ascontiguousarray(a.wavdata.data) for a in an]
    gmm = pl.synthetic_gmm(order=24, components=4, seed=2, n_frames=4000)
    wav, pcm, dwav, dpcm = corpus.convert_batch(waves_in, fs, gmm, pcm=True, diff=True)
    assert all(w is not None for w in wav) and all(p.dtype == torch.int16 for p in pcm)
    dg = pl.DeviceGMM(gmm.weights_, gmm.means_, gmm.covariances_, 'cuda:0')
    ctx = _lib.Context(0)
    for i, a in enumerate(an):
        mc = torch.from_numpy(np.ascontiguousarray(a.mel_cepstrum.data)).cuda()
        out = torch.empty_like(mc)
        torch.cuda.synchronize()
        _lib.check(ctx, lib.kwy_convert_mcep_dev(ctx.handle, c_vp(mc.data_ptr()), len(mc), 24, dg.M,
                                                 c_vp(dg.model(diff=True).data_ptr()), c_vp(out.data_ptr())))
        ctx.sync()
        rec = k.feature(a).mel_cepstrum
        rec.data = out.cpu().numpy()
        ref = k.apply_mlsa_filter(a.wavdata, rec)
        got = dwav[i].cpu().numpy()
        assert got.shape == ref.data.shape
        assert np.abs(got - ref.data).max() <= 1e-12 * max(1.0, np.abs(ref.data).max())
        assert np.array_equal(dpcm[i].cpu().numpy(), k.Wavdata(fs, got.copy())._pcm16(True, {}))


def test_lockstep_pairs_wav_in_pcm_out_and_its_graph():
    """PairBatchPipeline(wav_in=True, pcm=True), two waves: the f0 tracks extracted inside the step are the host
    entries', every waveform equals the pipeline that is GIVEN those tracks bit for bit, the int16 samples equal the
    host post-step of those waveforms, and the captured step (one HIP graph over four streams) replays to the same
    samples."""
    import torch
    from kwiiyatta_amd import pipeline as pl
    from kwiiyatta_amd.backend import world
    from kwiiyatta_amd.synthetic import make_utterance
    fs = 48000
    raw = [(make_utterance(seed=20 + 2 * i, fs=fs, seconds=1.0 + 0.1 * i)[0],
            make_utterance(seed=21 + 2 * i, fs=fs, seconds=1.2 + 0.05 * i, time_warp=1.1, formant_scale=1.1)[0]) for i in range(5)]
    gmm = pl.synthetic_gmm(order=24, components=4, seed=0, n_frames=3000)
    dg = pl.DeviceGMM(gmm.weights_, gmm.means_, gmm.covariances_, torch.device('cuda', 0))
    rng = np.random.default_rng(0)
    K = 1025
    silence = [np.abs(rng.normal(0, pl.EPS / fs, (pl.PAD_LEN, K))) for _ in range(4 * len(raw))]
    pw = pl.PairBatchPipeline(0, fs, raw, dg, waves=2, silence=silence, wav_in=True, pcm=True)
    pw.run(); pw.sync()
    assert not pw.f0_status().any()

    def track(x):
        f0c, t = world.dio(x, fs)
        return (x, world.stonemask(x, f0c, t, fs), t)
    given = [(track(a), track(b)) for a, b in raw]
    wv0 = pw.waves[0]
    assert np.array_equal(wv0.f0[0].cpu().numpy(), given[0][0][1]) and np.array_equal(wv0.t[1].cpu().numpy(), given[0][1][2])
    pg = pl.PairBatchPipeline(0, fs, given, dg, waves=2, silence=silence)
    pg.run(); pg.sync()
    first = []
    for k in range(len(raw)):
        a, b = pw.wave(k).cpu().numpy(), pg.wave(k).cpu().numpy()
        assert np.array_equal(a, b), k
        pcm = pw.pcm16(k).cpu().numpy()
        assert np.array_equal(pcm, _host_pcm(a, len(given[k][1][1]), fs)), k
        first.append(pcm)
    pw.capture()
    for k in range(len(raw)):
        pw.pcm16(k).zero_()
    pw.replay(); pw.sync()
    for k in range(len(raw)):
        assert np.array_equal(pw.pcm16(k).cpu().numpy(), first[k]), k
