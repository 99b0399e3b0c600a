"""GPU parity tests proper: the HIP WORLD kernels, called through the C ABI
(libkwy.so via ctypes), against the CPU oracle on identical inputs.

Tolerances (float64 path; north_star: output within 1e-4 RMS of the CPU path):
  * spectral envelope: max |d| / max|ref| per utterance <= 1e-8, and the
    log-spectral distance per bin <= 1e-3 on bins that are within 100 dB of the
    frame maximum, and sum|d| / sum|ref| <= 1e-9.  (CheapTrick's linear smoothing differences two cumulative
    sums; bins far below the frame's total power are conditioned like
    eps * total / local in ANY summation order, CPU included, so elementwise
    agreement on dead bins is not a meaningful target.)
  * aperiodicity: max abs <= 1e-4 (values live in [0, 1])
  * waveform: RMS <= 1e-9 given identical features (pulse positions, noise
    stream and overlap-add are reproduced exactly; only FFT rounding differs).
"""
import numpy as np
import pytest
from scipy.io import wavfile

from conftest import CLB_WAV, SLT_WAV, clb_variant

pytestmark = pytest.mark.gpu


def load(path):
    fs, d = wavfile.read(path)
    return fs, np.ascontiguousarray(d.astype(np.float64) / 2 ** 15)


@pytest.fixture(scope='module')
def ko():
    from oracle import oracle
    return oracle


@pytest.fixture(scope='module')
def kw():
    from kwiiyatta_amd.backend import world
    return world


def f0_track(ko, x, fs):
    f0, t = ko.dio(x, fs)
    return ko.stonemask(x, f0, t, fs), t


def check_spectrum(got, ref):
    assert got.shape == ref.shape
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() / np.abs(ref).max() <= 1e-8
    assert np.abs(got - ref).sum() / np.abs(ref).sum() <= 1e-9
    live = ref >= ref.max(axis=1, keepdims=True) * 1e-10
    lsd = np.abs(np.log(got[live]) - np.log(ref[live]))
    assert lsd.max() <= 1e-3, lsd.max()


CASES = [CLB_WAV, SLT_WAV, clb_variant('22'), clb_variant('44'), clb_variant('48'), clb_variant('96')]


@pytest.mark.parametrize('path', CASES)
def test_cheaptrick_parity(ko, kw, path):
    fs, x = load(path)
    f0, t = f0_track(ko, x, fs)
    check_spectrum(kw.cheaptrick(x, f0, t, fs), ko.cheaptrick(x, f0, t, fs))


@pytest.mark.parametrize('path', CASES)
def test_d4c_parity(ko, kw, path):
    fs, x = load(path)
    f0, t = f0_track(ko, x, fs)
    got, ref = kw.d4c(x, f0, t, fs), ko.d4c(x, f0, t, fs)
    assert got.shape == ref.shape
    # same frames pass the LoveTrain gate
    assert np.array_equal(got[:, 0] < 0.99, ref[:, 0] < 0.99)
    assert np.abs(got - ref).max() <= 1e-4
    assert np.median(np.abs(got - ref)) <= 1e-6


@pytest.mark.parametrize('path', [CLB_WAV, clb_variant('22'), clb_variant('48'), clb_variant('96')])
def test_synthesis_parity(ko, kw, path):
    fs, x = load(path)
    f0, t = f0_track(ko, x, fs)
    sp, ap = ko.cheaptrick(x, f0, t, fs), ko.d4c(x, f0, t, fs)
    got, ref = kw.synthesize(f0, sp, ap, fs, 5.0), ko.synthesize(f0, sp, ap, fs, 5.0)
    assert got.shape == ref.shape
    rms = np.sqrt(np.mean((got - ref) ** 2))
    assert rms <= 1e-9, rms
    assert np.abs(got - ref).max() <= 1e-8


def test_synthesis_leading_silence(ko, kw):
    """Recorded speech starts with silence: the excitation phase stays exactly 0 over those samples.  The phase
    scan must take such a run in one step per tile (it once took the samples one per workgroup round: ~0.6 us per
    silent sample), and the result must not change."""
    import time
    fs, x = load(clb_variant('48'))
    f0, t = f0_track(ko, x, fs)
    sp, ap = ko.cheaptrick(x, f0, t, fs), ko.d4c(x, f0, t, fs)
    lead = 1600                                         # 8 s of unvoiced frames in front
    f0s = np.r_[np.zeros(lead), f0]
    sps = np.ascontiguousarray(np.r_[np.repeat(sp[:1], lead, axis=0), sp])
    aps = np.ascontiguousarray(np.r_[np.repeat(ap[:1], lead, axis=0), ap])
    kw.synthesize(f0s, sps, aps, fs, 5.0)               # tables, arena
    t0 = time.perf_counter(); got = kw.synthesize(f0s, sps, aps, fs, 5.0); t_sil = time.perf_counter() - t0
    v = np.where(f0s > 0, f0s, 120.0)                   # the same length, voiced throughout
    t0 = time.perf_counter(); kw.synthesize(v, sps, aps, fs, 5.0); t_voiced = time.perf_counter() - t0
    ref = ko.synthesize(f0s, sps, aps, fs, 5.0)
    assert np.sqrt(np.mean((got - ref) ** 2)) <= 1e-9
    assert t_sil < t_voiced + 0.1, (t_sil, t_voiced)


def test_synthesis_plan_render_split(ko, kw):
    """kwy_synth_plan_dev + kwy_synth_render_dev (pulse placement from f0 alone, then the rendering) give the
    waveform of kwy_synthesize_dev bit for bit, also when the plan is made on another context / stream."""
    import torch
    from kwiiyatta_amd import _lib
    lib = _lib.lib
    fs, x = load(clb_variant('48'))
    f0, t = f0_track(ko, x, fs)
    sp, ap = kw.cheaptrick(x, f0, t, fs), kw.d4c(x, f0, t, fs)
    dev = torch.device('cuda', 0)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)    # noqa: E731
    df0, dsp, dap = d(f0), d(sp), d(ap)
    T, fft = len(f0), (sp.shape[1] - 1) * 2
    ylen = lib.kwy_synth_length(T, 5.0, fs)
    p = lambda a: _lib.c_vp(a.data_ptr())                              # noqa: E731
    main = _lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    side_stream = torch.cuda.Stream(device=dev)
    side = _lib.Context(0, stream=side_stream.cuda_stream)
    whole = torch.empty(ylen, dtype=torch.float64, device=dev)
    _lib.check(main, lib.kwy_synthesize_dev(main.handle, p(df0), T, p(dsp), p(dap), fft, 5.0, fs, 1.0, ylen, p(whole)))
    plan = torch.empty(lib.kwy_synth_plan_bytes(ylen), dtype=torch.uint8, device=dev)
    split = torch.empty(ylen, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    _lib.check(side, lib.kwy_synth_plan_dev(side.handle, p(df0), T, fft, 5.0, fs, ylen, p(plan)))
    side_stream.synchronize()
    _lib.check(main, lib.kwy_synth_render_dev(main.handle, p(plan), T, p(dsp), p(dap), fft, 5.0, fs, 1.0, ylen,
                                              p(split)))
    torch.cuda.synchronize()
    assert torch.equal(whole, split)
    assert np.array_equal(whole.cpu().numpy(), kw.synthesize(f0, sp, ap, fs, 5.0))


def test_synthesis_render_batch(ko, kw):
    """kwy_synth_render_batch_dev: the pulses of several utterances (different lengths, one of them unvoiced
    throughout) rendered by one pass of launches -- every waveform equals the single-utterance call's bit for bit."""
    import torch
    from kwiiyatta_amd import _lib
    lib = _lib.lib
    fs, x = load(clb_variant('48'))
    f0, t = f0_track(ko, x, fs)
    sp, ap = kw.cheaptrick(x, f0, t, fs), kw.d4c(x, f0, t, fs)
    dev = torch.device('cuda', 0)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)    # noqa: E731
    p = lambda a: _lib.c_vp(a.data_ptr())                              # noqa: E731
    ctx = _lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    fft = (sp.shape[1] - 1) * 2
    cuts = [(0, len(f0)), (40, 300), (100, 101 + 2), (0, 150)]
    jobs, want = [], []
    for n, (a, b) in enumerate(cuts):
        f = f0[a:b].copy()
        if n == 3:
            f[:] = 0.0
        T = len(f)
        ylen = lib.kwy_synth_length(T, 5.0, fs)
        df0, dsp, dap = d(f), d(sp[a:b]), d(ap[a:b])
        plan = torch.empty(lib.kwy_synth_plan_bytes(ylen), dtype=torch.uint8, device=dev)
        _lib.check(ctx, lib.kwy_synth_plan_dev(ctx.handle, p(df0), T, fft, 5.0, fs, ylen, p(plan)))
        single = torch.empty(ylen, dtype=torch.float64, device=dev)
        _lib.check(ctx, lib.kwy_synth_render_dev(ctx.handle, p(plan), T, p(dsp), p(dap), fft, 5.0, fs, 1.0, ylen,
                                                 p(single)))
        torch.cuda.synchronize()        # (the context has its own stream: torch's copies do not wait for it)
        want.append(single.cpu().numpy())
        jobs.append((plan, dsp, dap, torch.full((ylen,), np.nan, dtype=torch.float64, device=dev)))
    torch.cuda.synchronize()
    arr = _lib.synth_job_array(jobs)
    _lib.check(ctx, lib.kwy_synth_render_batch_dev(ctx.handle, arr, len(jobs), fft, 5.0, fs, 1.0))
    torch.cuda.synchronize()
    for (plan, dsp, dap, y), w in zip(jobs, want):
        assert np.array_equal(y.cpu().numpy(), w)
    assert np.array_equal(want[0], kw.synthesize(f0, sp, ap, fs, 5.0))


def test_synthesis_voicing_patterns(ko, kw):
    """Random voiced / unvoiced schedules -- silence at the start, in the middle, at the end, single voiced frames,
    f0 jumps -- at two rates: the exact-rounding phase scan takes different routes through its tiles (fast tiles,
    binade changes, runs of zero increments), the waveform must not notice."""
    rng = np.random.default_rng(77)
    for case, fs in enumerate((16000, 16000, 48000, 16000, 48000, 16000)):
        T = int(rng.integers(40, 400))
        K = kw.get_cheaptrick_fft_size(fs) // 2 + 1
        f0 = np.zeros(T)
        t = 0
        while t < T:
            run = int(rng.integers(1, 60))
            if rng.random() < 0.55:
                f0[t:t + run] = rng.uniform(60, 600) * np.exp(0.2 * np.sin(np.arange(min(run, T - t)) / 7.0))
            t += run
        if case == 0:
            f0[:] = 0.0                                  # nothing voiced at all
        if case == 1:
            f0[:30] = 0.0; f0[-25:] = 0.0                 # silence at both ends
        k = np.arange(K)
        sp = np.exp(-k[None, :] / (K / 6.0)) * (1.0 + 0.3 * rng.random((T, 1))) * 1e-3 + 1e-9
        ap = np.clip(0.1 + 0.8 * k[None, :] / K + 0.05 * rng.standard_normal((T, K)), 0.001, 0.999)
        sp, ap = np.ascontiguousarray(sp), np.ascontiguousarray(ap)
        got, ref = kw.synthesize(f0, sp, ap, fs, 5.0), ko.synthesize(f0, sp, ap, fs, 5.0)
        assert got.shape == ref.shape
        scale = max(np.sqrt(np.mean(ref ** 2)), 1e-12)
        assert np.sqrt(np.mean((got - ref) ** 2)) <= 1e-9 * max(scale, 1.0) + 1e-12 * scale, (case, fs, T)
        assert np.array_equal(got, kw.synthesize(f0, sp, ap, fs, 5.0))


@pytest.mark.parametrize('fs,seconds,f0_hz,lead', [
    (16000, 30.0, 1250.0, 0),      # large increments (f0 < fs / 12): the first tiles cross several binades each, 18 binades in all
    (48000, 20.0, 55.0, 0),        # small increments: long runs of fast tiles between the crossings
    (16000, 12.0, 1200.0, 700),    # 3.5 s of silence in front, then a steep start in the middle of a tile
    (48000, 6.0, 0.0, 0),          # nothing voiced: the 500 Hz default throughout
])
def test_synthesis_phase_chain_extremes(ko, kw, fs, seconds, f0_hz, lead):
    """The exact-rounding phase chain on inputs that push it through its rarer routes: tiles with more than one
    binade change (the chunk walk gives up and the workgroup rounds take over), many slow tiles, a warm-up that
    starts inside a tile, segments handed to the parallel pass.  Pulse placement decides the waveform: it must equal
    the oracle's."""
    rng = np.random.default_rng(int(fs + f0_hz))
    T = int(seconds * 200)
    K = kw.get_cheaptrick_fft_size(fs) // 2 + 1
    f0 = np.full(T, f0_hz) * np.exp(0.05 * np.sin(np.arange(T) / 23.0)) if f0_hz > 0 else np.zeros(T)
    f0[:lead] = 0.0
    if f0_hz > 0:
        f0[T // 2:T // 2 + 40] = 0.0                      # an unvoiced stretch in the middle
    k = np.arange(K)
    sp = np.ascontiguousarray(np.exp(-k[None, :] / (K / 6.0)) * (1.0 + 0.3 * rng.random((T, 1))) * 1e-3 + 1e-9)
    ap = np.ascontiguousarray(np.clip(0.1 + 0.8 * k[None, :] / K + np.zeros((T, 1)), 0.001, 0.999))
    got, ref = kw.synthesize(f0, sp, ap, fs, 5.0), ko.synthesize(f0, sp, ap, fs, 5.0)
    assert got.shape == ref.shape
    scale = max(np.sqrt(np.mean(ref ** 2)), 1e-12)
    assert np.sqrt(np.mean((got - ref) ** 2)) <= 1e-9 * max(scale, 1.0) + 1e-12 * scale
    assert np.array_equal(got, kw.synthesize(f0, sp, ap, fs, 5.0))


def test_synthesis_is_deterministic(ko, kw):
    """Two runs give the same bits (ordered overlap-add, no floating-point atomics): what the reference asserts
    with `(analyzer_wav.data == feature_wav.data).all()`, tests/kwiiyatta/test_vocoder.py:171."""
    fs, x = load(clb_variant('48'))
    f0, t = f0_track(ko, x, fs)
    sp, ap = kw.cheaptrick(x, f0, t, fs), kw.d4c(x, f0, t, fs)
    first = kw.synthesize(f0, sp, ap, fs, 5.0)
    for _ in range(3):
        assert np.array_equal(kw.synthesize(f0, sp, ap, fs, 5.0), first)
    # many more pulses than items per tile: f0 near the upper limit
    f0h = np.full_like(f0, 3000.0)
    a = kw.synthesize(f0h, sp, ap, fs, 5.0)
    assert np.array_equal(a, kw.synthesize(f0h, sp, ap, fs, 5.0))
    assert np.sqrt(np.mean((a - ko.synthesize(f0h, sp, ap, fs, 5.0)) ** 2)) <= 1e-9


def test_analysis_synthesis_end_to_end_rms(ko, kw):
    """GPU analyse -> GPU synthesise vs oracle analyse -> oracle synthesise:
    the north-star criterion (waveform within 1e-4 RMS)."""
    fs, x = load(clb_variant('48'))
    f0, t = f0_track(ko, x, fs)
    y_gpu = kw.synthesize(f0, kw.cheaptrick(x, f0, t, fs), kw.d4c(x, f0, t, fs), fs, 5.0)
    y_ref = ko.synthesize(f0, ko.cheaptrick(x, f0, t, fs), ko.d4c(x, f0, t, fs), fs, 5.0)
    rms = np.sqrt(np.mean((y_gpu - y_ref) ** 2))
    assert rms <= 1e-4, rms
    assert rms <= 1e-6 * np.sqrt(np.mean(y_ref ** 2)) * 1e3


@pytest.mark.parametrize('frame_period', [3, 5, 8])
def test_frame_periods(ko, kw, frame_period):
    fs, x = load(CLB_WAV)
    f0, t = ko.dio(x, fs, frame_period=frame_period)
    f0 = ko.stonemask(x, f0, t, fs)
    check_spectrum(kw.cheaptrick(x, f0, t, fs), ko.cheaptrick(x, f0, t, fs))
    sp, ap = ko.cheaptrick(x, f0, t, fs), ko.d4c(x, f0, t, fs)
    got = kw.synthesize(f0, sp, ap, fs, float(frame_period))
    ref = ko.synthesize(f0, sp, ap, fs, float(frame_period))
    assert np.sqrt(np.mean((got - ref) ** 2)) <= 1e-9


@pytest.mark.parametrize('fs,up,down', [(8000, 1, 2), (12000, 3, 4)])
def test_low_sampling_rates(ko, kw, fs, up, down):
    """Rates below the reference's fixtures: other FFT sizes in every kernel (D4C 1024, CheapTrick 512),
    no / one aperiodicity band, and LoveTrain's 7.9 kHz boundary above Nyquist (undefined upstream,
    zero-extended in the oracle)."""
    import scipy.signal as ss
    _, x16 = load(CLB_WAV)
    x = np.ascontiguousarray(ss.resample_poly(x16, up, down))
    f0, t = f0_track(ko, x, fs)
    check_spectrum(kw.cheaptrick(x, f0, t, fs), ko.cheaptrick(x, f0, t, fs))
    got, ref = kw.d4c(x, f0, t, fs), ko.d4c(x, f0, t, fs)
    assert np.array_equal(got[:, 0] < 0.99, ref[:, 0] < 0.99)
    assert np.abs(got - ref).max() <= 1e-4
    sp, ap = ko.cheaptrick(x, f0, t, fs), ref
    assert np.sqrt(np.mean((kw.synthesize(f0, sp, ap, fs) - ko.synthesize(f0, sp, ap, fs)) ** 2)) <= 1e-9


def test_options(ko, kw):
    """q1 / fft_size / threshold keyword paths (reference call sites
    tests/kwiiyatta/test_vocoder.py:429-430, view/qt/kwiieiya.py:84)."""
    fs, x = load(CLB_WAV)
    f0, t = f0_track(ko, x, fs)
    check_spectrum(kw.cheaptrick(x, f0, t, fs, q1=-0.09, fft_size=2048),
                   ko.cheaptrick(x, f0, t, fs, q1=-0.09, fft_size=2048))
    for thr in (0.0, 0.5, 0.95):
        got, ref = kw.d4c(x, f0, t, fs, threshold=thr), ko.d4c(x, f0, t, fs, threshold=thr)
        assert np.abs(got - ref).max() <= 1e-4
    got, ref = kw.d4c(x, f0, t, fs, fft_size=2048), ko.d4c(x, f0, t, fs, fft_size=2048)
    assert got.shape == (len(f0), 1025) and np.abs(got - ref).max() <= 1e-4


def test_edge_inputs(ko, kw):
    fs = 16000
    rng = np.random.default_rng(0)
    # all-unvoiced, digital silence, one-frame and clipped-window inputs
    x = rng.standard_normal(4000) * 0.01
    f0 = np.zeros(51)
    t = np.arange(51) * 0.005
    check_spectrum(kw.cheaptrick(x, f0, t, fs), ko.cheaptrick(x, f0, t, fs))
    assert np.array_equal(kw.d4c(x, f0, t, fs), ko.d4c(x, f0, t, fs))
    xs = np.zeros(4000)
    got, ref = kw.cheaptrick(xs, f0, t, fs), ko.cheaptrick(xs, f0, t, fs)
    assert np.abs(np.log(got) - np.log(ref)).max() <= 1e-6   # spectrum of the noise floor itself
    sp = ko.cheaptrick(x, f0, t, fs)
    ap = ko.d4c(x, f0, t, fs)
    assert np.sqrt(np.mean((kw.synthesize(f0, sp, ap, fs) - ko.synthesize(f0, sp, ap, fs)) ** 2)) <= 1e-12
    # very short signal, frames beyond the end of x
    x1 = rng.standard_normal(90) * 0.1
    f01 = np.array([0.0, 120.0, 0.0]); t1 = np.arange(3) * 0.005
    check_spectrum(kw.cheaptrick(x1, f01, t1, fs), ko.cheaptrick(x1, f01, t1, fs))


def test_contiguity_contract(kw):
    """pyworld raises ValueError('ndarray is not C-contiguous')
    (reference tests/kwiiyatta/vocoder/test_world.py:21-40)."""
    x = np.zeros(2000)
    f0 = np.zeros(10); t = np.arange(10) * 0.005
    with pytest.raises(ValueError) as e:
        kw.cheaptrick(x[::2], f0, t, 16000)
    assert str(e.value) == 'ndarray is not C-contiguous'
    with pytest.raises(ValueError) as e:
        kw.d4c(x[::2], f0, t, 16000)
    assert str(e.value) == 'ndarray is not C-contiguous'
    sp = np.ones((10, 513)); ap = np.ones((10, 513))
    with pytest.raises(ValueError) as e:
        kw.synthesize(f0[::2], sp[::2], ap[::2], 16000)
    assert str(e.value) == 'ndarray is not C-contiguous'


def test_full_size_config2(ko, kw):
    """BASELINE config 2: one 48 kHz / 10 s synthetic utterance (T=2001, K=1025),
    analyse + synthesise, against the oracle at full size."""
    from kwiiyatta_amd.synthetic import make_utterance
    fs = 48000
    x, f0, t = make_utterance(seed=1234, fs=fs, seconds=10.0)
    assert len(f0) == 2001
    sp, ap = kw.cheaptrick(x, f0, t, fs), kw.d4c(x, f0, t, fs)
    assert sp.shape == (2001, 1025) and ap.shape == (2001, 1025)
    sp_ref, ap_ref = ko.cheaptrick(x, f0, t, fs), ko.d4c(x, f0, t, fs)
    check_spectrum(sp, sp_ref)
    assert np.abs(ap - ap_ref).max() <= 1e-4
    y = kw.synthesize(f0, sp, ap, fs, 5.0)
    assert len(y) == 480240
    y_ref = ko.synthesize(f0, sp_ref, ap_ref, fs, 5.0)
    rms = np.sqrt(np.mean((y - y_ref) ** 2))
    assert rms <= 1e-4, rms
    # size-independent properties: amplitude scaling and determinism
    sp2 = kw.cheaptrick(x * 0.5, f0, t, fs)
    live = sp >= sp.max(axis=1, keepdims=True) * 1e-10
    assert np.abs(sp2[live] / sp[live] - 0.25).max() <= 1e-6
    assert np.array_equal(sp, kw.cheaptrick(x, f0, t, fs))
    assert ((ap > 0) & (ap <= 1)).all()


def test_synthesis_8192_points(ko, kw):
    """fft_size 8192 (K = 4097): what the reference synthesises after resampling features up to 96 kHz
    (3078 bins are reshaped to 4097, kwiiyatta/vocoder/world.py:71-78)."""
    from scipy.interpolate import interp1d
    fs, x = load(clb_variant('96'))
    f0, t = f0_track(ko, x, fs)
    f0, t = np.ascontiguousarray(f0[:120]), t[:120]
    sp, ap = ko.cheaptrick(x, f0, t, fs), ko.d4c(x, f0, t, fs)
    grid, fine = np.linspace(0, 1, sp.shape[1]), np.linspace(0, 1, 4097)
    sp8 = np.ascontiguousarray(np.exp(interp1d(grid, np.log(sp), axis=1)(fine)))
    ap8 = np.ascontiguousarray(np.clip(interp1d(grid, ap, axis=1)(fine), 1e-3, 1 - 1e-12))
    got, ref = kw.synthesize(f0, sp8, ap8, fs, 5.0), ko.synthesize(f0, sp8, ap8, fs, 5.0)
    assert got.shape == ref.shape
    assert np.sqrt(np.mean((got - ref) ** 2)) <= 1e-9 * max(1.0, np.abs(ref).max())


# ---------------------------------------------------------------- the randn table and the jump-ahead path beyond it
def test_randn_table_equals_serial_stream(ko):
    """WORLD's randn stream is a constant; the device keeps its first 2^L draws in a table.  Beginning, middle (across
    the 4096-draw chunks the fill kernel works in) and end of the table against the oracle's serial generator."""
    from kwiiyatta_amd import _lib
    ctx = _lib.Context(0)
    n = ctx.set_randn_limit(-1)
    assert n >= 1 << 12
    m = min(n, 3_000_000)
    ref = ko.randn(m)
    got = np.empty(m)
    _lib.check(ctx, _lib.lib.kwy_randn_stream(ctx.handle, 0, m, _lib.ptr(got)))
    assert np.array_equal(got, ref)
    with pytest.raises(ValueError):
        _lib.check(ctx, _lib.lib.kwy_randn_stream(ctx.handle, n - 4, 8, _lib.ptr(got)))


@pytest.mark.parametrize('path,limit', [(clb_variant('48'), 0), (clb_variant('48'), 30_000), (clb_variant('48'), 150_000),
                                        (clb_variant('48'), 400_000), (CLB_WAV, 77_777), (clb_variant('96'), 300_000)])
def test_jump_ahead_beyond_the_table_gives_the_same_bits(ko, kw, path, limit):
    """With the table cut short, the frames / pulses whose draws lie beyond it compute them by GF(2) jump-ahead inside
    the kernels: every output must equal the all-table result bit for bit (limit 0: no table at all; the other limits
    put the switch in the middle of CheapTrick, of the LoveTrain pass, of the D4C body and of the synthesis)."""
    from kwiiyatta_amd import _lib
    fs, x = load(path)
    x = x[:int(1.2 * fs)]
    f0, t = f0_track(ko, x, fs)
    full, cut = _lib.Context(0), _lib.Context(0)
    assert cut.set_randn_limit(limit) == limit
    out = []
    for ctx in (full, cut):
        sp = kw.cheaptrick(x, f0, t, fs, ctx=ctx)
        ap = kw.d4c(x, f0, t, fs, ctx=ctx)
        y = kw.synthesize(f0, sp, ap, fs, 5.0, ctx=ctx)
        out.append((sp, ap, y))
    for a, b in zip(*out):
        assert np.array_equal(a, b)
    check_spectrum(out[1][0], ko.cheaptrick(x, f0, t, fs))
    assert np.abs(out[1][1] - ko.d4c(x, f0, t, fs)).max() <= 1e-4


@pytest.mark.parametrize('path', [clb_variant('48'), CLB_WAV, clb_variant('96')])
def test_batched_analysis_equals_single_calls(ko, kw, path):
    """kwy_cheaptrick_batch_dev / kwy_d4c_batch_dev: utterances of different lengths in one grid per kernel (more of
    them than one launch takes: 19 > KWY_BATCH_MAX = 16) give bit for bit what the single-utterance calls give --
    every utterance's noise stream starts at draw 0, as every pyworld call reseeds."""
    import torch
    from kwiiyatta_amd import _lib
    from kwiiyatta_amd._lib import lib
    fs, x = load(path)
    ctx = _lib.Context(0)
    utts = []
    for k in range(19):
        xs = np.ascontiguousarray(x[int(0.03 * k * fs):int((0.45 + 0.05 * (k % 5)) * fs) + int(0.03 * k * fs)])
        f0, t = f0_track(ko, xs, fs)
        utts.append((xs, f0, t))
    fft = lib.kwy_cheaptrick_fft_size(fs, 71.0)
    K = fft // 2 + 1
    dev = [tuple(torch.from_numpy(a).cuda() for a in u) for u in utts]
    sp = [torch.empty((len(u[1]), K), dtype=torch.float64, device='cuda') for u in utts]
    ap = [torch.empty((len(u[1]), K), dtype=torch.float64, device='cuda') for u in utts]
    torch.cuda.synchronize()
    arr = _lib.utterance_array([(d[0], d[2], d[1], o) for d, o in zip(dev, sp)])
    _lib.check(ctx, lib.kwy_cheaptrick_batch_dev(ctx.handle, arr, len(utts), fs, -0.15, 71.0, fft, 1.0))
    arr = _lib.utterance_array([(d[0], d[2], d[1], o) for d, o in zip(dev, ap)])
    _lib.check(ctx, lib.kwy_d4c_batch_dev(ctx.handle, arr, len(utts), fs, 0.85, fft))
    ctx.sync()
    for (xs, f0, t), s_, a_ in zip(utts, sp, ap):
        assert np.array_equal(s_.cpu().numpy(), kw.cheaptrick(xs, f0, t, fs))
        assert np.array_equal(a_.cpu().numpy(), kw.d4c(xs, f0, t, fs))
