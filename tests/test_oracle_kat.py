"""Pins the CPU oracle (oracle/) against the reference's own known-answer
envelopes -- the only numbers the reference's tests hold for this path
(2 significant digits, through the whole analyse/synthesise stack on the CMU
ARCTIC fixtures).  No GPU needed."""
import numpy as np
import pytest
from scipy.io import wavfile

from conftest import CLB_WAV, CLB_WAV2, round_equal
from oracle import oracle as ko
from refmetrics import calc_powered_diff, feature_diffs


def load(path):
    fs, d = wavfile.read(path)
    assert d.dtype == np.int16
    return fs, np.ascontiguousarray(d.astype(np.float64) / 2 ** 15)


def analyze(x, fs, frame_period=5, order=24):
    """kwiiyatta WorldAnalyzer semantics (reference kwiiyatta/vocoder/world.py:33-59)."""
    f0, t = ko.dio(x, fs, frame_period=frame_period)
    f0 = ko.stonemask(x, f0, t, fs)
    sp = ko.cheaptrick(x, f0, t, fs) / fs
    ap = ko.d4c(x, f0, t, fs)
    mc = ko.sp2mc(sp, order, ko.mcepalpha(fs))
    return dict(fs=fs, f0=f0, t=t, sp=sp, ap=ap, mc=mc)


def normalize_data(d, peak_lv=-1):
    peak = np.abs(d).max()
    mx = np.power(10, peak_lv / 10)
    if peak > mx:
        d *= mx / peak


def synth_and_save(a, sp=None):
    """Feature.synthesize() post-processing + Wavdata.save() int16 round trip
    (reference vocoder/abc/synthesizer.py:11-20, wavfile.py:20-29)."""
    sp = a['sp'] if sp is None else sp
    y = ko.synthesize(a['f0'], np.ascontiguousarray(sp * a['fs']), a['ap'], a['fs'], 5.0)
    y = y - y.mean()
    fs = a['fs']
    for i in range(len(a['f0'])):
        normalize_data(y[fs * i // 1000: fs * (i + 1) // 1000])
    y = y - y.mean()
    normalize_data(y)
    return (y * 2 ** 15).astype(np.int16).astype(np.float64) / 2 ** 15


@pytest.fixture(scope='module')
def clb1():
    fs, x = load(CLB_WAV)
    return analyze(x, fs)


def test_mcepalpha_values():
    # pysptk.util.mcepalpha; values quoted in SURVEY.md section 2 (N8)
    assert abs(ko.mcepalpha(16000) - 0.41) < 1e-9
    assert abs(ko.mcepalpha(48000) - 0.554) < 1e-9


def test_sizes():
    assert ko.get_cheaptrick_fft_size(16000) == 1024
    assert ko.get_cheaptrick_fft_size(48000) == 2048
    assert ko.get_cheaptrick_fft_size(96000) == 4096


def test_analyze_difffile(clb1):
    """reference tests/kwiiyatta/test_vocoder.py:140-149: 0.63 / 1.0 / 0.49 / 0.27"""
    fs, x2 = load(CLB_WAV2)
    a2 = analyze(x2, fs)
    f0d, spd, apd, mcd = feature_diffs(clb1, a2, strict=False)
    assert round_equal(0.63, f0d), f0d
    assert round_equal(1.0, spd), spd
    assert round_equal(0.49, apd), apd
    assert round_equal(0.27, mcd), mcd


def test_voice_resynthesis(clb1):
    """reference tests/kwiiyatta/test_resynthesize_voice.py:19-39: 0.079/0.20/0.073/0.054"""
    b = analyze(synth_and_save(clb1), clb1['fs'])
    f0d, spd, apd, mcd = feature_diffs(clb1, b)
    assert round_equal(0.079, f0d), f0d
    assert round_equal(0.20, spd), spd
    assert round_equal(0.073, apd), apd
    assert round_equal(0.054, mcd), mcd


def test_voice_resynthesis_mcep(clb1):
    """reference tests/kwiiyatta/test_resynthesize_voice.py:42-62: 0.081/0.22/0.087/0.051"""
    sp_m = ko.mc2sp(clb1['mc'], ko.mcepalpha(clb1['fs']), 1024)
    b = analyze(synth_and_save(clb1, sp_m), clb1['fs'])
    f0d, spd, apd, mcd = feature_diffs(clb1, b)
    assert round_equal(0.081, f0d), f0d
    assert round_equal(0.22, spd), spd
    assert round_equal(0.087, apd), apd
    assert round_equal(0.051, mcd), mcd


@pytest.mark.parametrize('order', [24, 36, 48])
def test_mcep_to_spec(order):
    """reference tests/kwiiyatta/test_vocoder.py:233-244: envelope (0.021, 0.091)"""
    fs, x = load(CLB_WAV)
    a = analyze(x, fs, order=order)
    sp_m = ko.mc2sp(a['mc'], ko.mcepalpha(fs), 1024)
    d = calc_powered_diff(a['sp'], sp_m)
    assert 0.021 < d < 0.091, d


def test_silence_synthesis_peak():
    """reference tests/kwiiyatta/test_vocoder.py:470-479: peak in (4e-8, 6e-8)"""
    for fs in (16000, 48000):
        K = ko.get_cheaptrick_fft_size(fs) // 2 + 1
        rng = np.random.RandomState(0)
        eps = 2.220446049250313e-16
        f0 = np.zeros(100)
        sp = np.abs(rng.normal(0, eps / fs, (100, K)))
        ap = np.full((100, K), 1 - 1e-12)
        y = ko.synthesize(f0, np.ascontiguousarray(sp * fs), ap, fs, 5.0)
        y = y - y.mean()
        assert 3.5e-8 < y.max() < 7e-8, y.max()


def test_chain_projection_equals_package_projection():
    """oracle/chain.py restates the reference's project_path_iter for the all-oracle config-3 chain; it must agree with
    the package's host logic on DTW-shaped paths and on paths with jumps in y."""
    import numpy as np
    from oracle import chain
    from kwiiyatta_amd.vocoder.align import project_path_iter
    rng = np.random.default_rng(3)
    for case in range(200):
        nx, ny = rng.integers(5, 60, 2)
        x = y = 0
        path = [(0, 0)]
        while x < nx - 1 or y < ny - 1:
            step = rng.integers(0, 3)
            jump = 1 if case % 3 else int(rng.integers(1, 4))
            if step == 0 and x < nx - 1:
                x += 1
            elif step == 1 and y < ny - 1:
                y = min(ny - 1, y + jump)
            else:
                x, y = min(nx - 1, x + 1), min(ny - 1, y + jump)
            path.append((x, y))
        for trim_len in (0, 1, 3):
            if ny <= 2 * trim_len + 1:
                continue
            want = list(project_path_iter(np.array(path), trim=True, trim_len=trim_len))
            assert chain.project_path(path, trim_len) == want
