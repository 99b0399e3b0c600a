"""The host package (kwiiyatta_amd) behind the reference's API: laziness and
caching semantics, sizes, the reference's known-answer envelopes and the two
CLIs.  Every test runs twice: with the CPU oracle injected behind the backend
modules (host logic, no GPU) and, under `-m gpu`, on the HIP kernels.

Semantics and numbers are those pinned by the reference's own tests
(/root/reference/tests/kwiiyatta/test_vocoder.py, test_resynthesize_voice.py,
test_convert_voice.py, test_converter.py, test_dataset.py)."""
import copy
import pathlib
import shutil
import sys

import numpy as np
import pytest

from conftest import CLB_DIR, CLB_WAV, CLB_WAV2, SLT_DIR, SLT_WAV, clb_variant, round_equal
from refmetrics import calc_diff, calc_powered_diff


def feature_diffs(exp, act, **kw):
    return (calc_diff(exp.f0, act.f0, **kw),
            calc_powered_diff(exp.spectrum_envelope, act.spectrum_envelope, **kw),
            calc_diff(exp.aperiodicity, act.aperiodicity, **kw),
            calc_diff(exp.mel_cepstrum.data, act.mel_cepstrum.data, **kw))


def test_lazy_extraction_and_caching(kwiiyatta):
    a = kwiiyatta.analyze_wav(CLB_WAV)
    assert a._f0 is None and a._spectrum_envelope is None and a._aperiodicity is None
    assert a._mel_cepstrum.data is None and a._is_voiced is None

    _ = a.aperiodicity                      # needs f0, nothing else
    assert a._f0 is not None and a._aperiodicity is not None
    assert a._spectrum_envelope is None and a._mel_cepstrum.data is None and a._is_voiced is None

    a._aperiodicity = None
    _ = a.mel_cepstrum                      # needs the spectrum, not the aperiodicity
    assert a._spectrum_envelope is not None and a._mel_cepstrum.data is not None
    assert a._aperiodicity is None and a._is_voiced is None

    a = kwiiyatta.analyze_wav(CLB_WAV)
    _ = a.is_voiced                         # f0 + aperiodicity only
    assert a._f0 is not None and a._aperiodicity is not None and a._is_voiced is not None
    assert a._spectrum_envelope is None and a._mel_cepstrum.data is None

    a = kwiiyatta.analyze_wav(CLB_WAV)
    _ = a.mel_cepstrum
    f = kwiiyatta.feature(a)                # materialises, shares the arrays
    assert f is not a and f.mel_cepstrum_order == a.mel_cepstrum_order
    assert a._f0 is f.f0 and a._spectrum_envelope is f._spectrum_envelope
    assert a._aperiodicity is f.aperiodicity
    assert a._mel_cepstrum.data is f._mel_cepstrum.data
    assert a._is_voiced is None
    assert (a.is_voiced == f.is_voiced).all()

    f = f[::2]
    g = copy.copy(f)
    assert not f.f0.flags['C_CONTIGUOUS'] and not f.spectrum_envelope.flags['C_CONTIGUOUS']
    f.ascontiguousarray()
    assert f.f0.flags['C_CONTIGUOUS'] and f.spectrum_envelope.flags['C_CONTIGUOUS']
    assert f.aperiodicity.flags['C_CONTIGUOUS']
    assert f is not g and f == g

    f = kwiiyatta.feature(a, mcep_order=a.mel_cepstrum_order * 2)
    assert f.mel_cepstrum_order != a.mel_cepstrum_order
    assert a._mel_cepstrum.data is not f.mel_cepstrum.data
    # a lower order is the prefix of a higher one
    assert np.allclose(a._mel_cepstrum.data, f.mel_cepstrum.data[:, :a.mel_cepstrum_order + 1],
                       rtol=0, atol=1e-9)
    f.mel_cepstrum_order = a.mel_cepstrum_order
    assert (a._mel_cepstrum.data == f.mel_cepstrum.data).all()

    f.f0 = None
    assert a.f0 is not None
    f.spectrum_envelope = None
    assert a.spectrum_envelope is not None


@pytest.mark.parametrize('suffix,fs', [(None, 16000), ('22', 22050), ('48', 48000)])
@pytest.mark.parametrize('frame_period', [3, 5, 8])
def test_analyzer_sizes(kwiiyatta, suffix, fs, frame_period):
    path = CLB_WAV if suffix is None else clb_variant(suffix)
    a = kwiiyatta.analyze_wav(path, frame_period=frame_period, mcep_order=36)
    assert a.fs == fs and a.mel_cepstrum_order == 36 and a.frame_period == frame_period
    assert a.spectrum_len == kwiiyatta.Synthesizer.fs_spectrum_len(fs)
    n = a.data.shape[0]
    assert a.frame_len == n * 1000 // fs // frame_period + 1
    f = kwiiyatta.feature(a)
    assert f.frame_len == a.frame_len == len(a.f0)
    assert f.spectrum_len == a.spectrum_len == a.spectrum_envelope.shape[1]
    assert calc_diff(a.f0, f.f0) == 0
    assert f._mel_cepstrum.data is None
    assert calc_diff(a.mel_cepstrum.data, f.mel_cepstrum.data) == 0
    assert a.mel_cepstrum.data.shape == (a.frame_len, 37)


def test_feature_equality_and_slicing(kwiiyatta):
    a = kwiiyatta.analyze_wav(CLB_WAV)
    f = kwiiyatta.feature(a)
    assert f == a
    f._mel_cepstrum._fs *= 2
    assert f != a
    f._mel_cepstrum._fs = a.fs
    f.spectrum_envelope = copy.copy(a.spectrum_envelope)
    assert f == a
    f.spectrum_envelope[0][0] += 0.001
    assert f != a
    f.spectrum_envelope[0][0] = a.spectrum_envelope[0][0]
    assert f == a
    half = len(a.f0) // 2
    f0, spec, ape, mcep = a[half]
    assert f0 == a.f0[half] and (spec == a.spectrum_envelope[half]).all()
    assert (ape == a.aperiodicity[half]).all() and (mcep == a.mel_cepstrum.data[half]).all()
    h = a[:half]
    assert len(h.f0) == len(h.spectrum_envelope) == len(h.aperiodicity) == half
    assert (h.mel_cepstrum.data == a.mel_cepstrum.data[:half]).all()
    with pytest.raises(TypeError):
        kwiiyatta.feature('x')
    with pytest.raises(TypeError):
        kwiiyatta.align(a, 3)
    with pytest.raises(TypeError):
        kwiiyatta.resample(3, 16000)


def test_analyze_difffile_kat(kwiiyatta):
    """test_vocoder.py:140-149"""
    a1, a2 = kwiiyatta.analyze_wav(CLB_WAV), kwiiyatta.analyze_wav(CLB_WAV2)
    f0d, spd, apd, mcd = feature_diffs(a1, a2, strict=False)
    assert round_equal(0.63, f0d) and round_equal(1.0, spd)
    assert round_equal(0.49, apd) and round_equal(0.27, mcd)


@pytest.mark.parametrize('frame_period', [3, 5, 8])
def test_reanalyze_kat(kwiiyatta, frame_period):
    """test_vocoder.py:152-181 envelope (i16 / 16 kHz column)"""
    a1 = kwiiyatta.analyze_wav(CLB_WAV, frame_period=frame_period)
    wav_a = a1.synthesize()
    wav_f = kwiiyatta.feature(a1).synthesize()
    assert wav_a.fs == wav_f.fs
    assert (wav_a.data == wav_f.data).all()      # exact, as the reference asserts (test_vocoder.py:171)
    a2 = kwiiyatta.Analyzer(wav_a, frame_period=frame_period)
    f0d, spd, apd, mcd = feature_diffs(a1, a2)
    assert 0.052 < f0d < 0.094 and 0.20 < spd < 0.22
    assert 0.063 < apd < 0.096 and 0.030 < mcd < 0.055


def test_silence_kat(kwiiyatta):
    """test_vocoder.py:470-479"""
    for fs in (16000, 48000):
        s = kwiiyatta.Synthesizer.create_silence_feature(100, fs)
        assert s.frame_len == 100
        assert 3.5e-8 < s.synthesize().data.max() < 7e-8


def _dtw_aligner(ko, m1, m2):
    """nnmnkwii.preprocessing.alignment.DTWAligner(verbose=0).transform, restated
    (trim zero frames, fastdtw radius 1, gather along the path, zero-pad)."""
    from kwiiyatta_amd.converter.dataset import trim_zeros_frames
    x, y = trim_zeros_frames(m1), trim_zeros_frames(m2)
    _, path = ko.fastdtw(x, y, radius=1, dist=2)
    px, py = [p[0] for p in path], [p[1] for p in path]
    x, y = x[px], y[py]
    n = max(len(m1), len(m2), len(x))
    X, Y = np.zeros((n, m1.shape[1])), np.zeros((n, m1.shape[1]))
    X[:len(x)], Y[:len(y)] = x, y
    return X, Y


def test_align_even_raw_equals_dtwaligner(kwiiyatta):
    """test_vocoder.py:247-263: exact equality with the plain DTW aligner"""
    from oracle import oracle as ko
    a1, a2 = kwiiyatta.analyze_wav(CLB_WAV), kwiiyatta.analyze_wav(SLT_WAV)
    # the aligner of the test uses radius 1
    act1, act2 = kwiiyatta.align_even(a1, a2, vuv=None, power='raw', strict=False, pad_silence=False,
                                      radius=1)
    exp1, exp2 = _dtw_aligner(ko, a1.mel_cepstrum.data, a2.mel_cepstrum.data)
    n = len(act1.mel_cepstrum.data)
    assert (exp1[:n] == act1.mel_cepstrum.data).all() and not exp1[n:].any()
    assert (exp2[:n] == act2.mel_cepstrum.data).all()


def test_align_even_kat(kwiiyatta):
    """test_vocoder.py:266-288: fastdtw distances 252 / 302 under np.random.seed(0)"""
    from oracle import oracle as ko
    np.random.seed(0)
    a1, a2 = kwiiyatta.analyze_wav(CLB_WAV), kwiiyatta.analyze_wav(SLT_WAV)
    exp1, exp2 = _dtw_aligner(ko, a1.mel_cepstrum.data, a2.mel_cepstrum.data)
    act1, act2 = kwiiyatta.align_even(a1, a2)
    d1, _ = ko.fastdtw(exp1, act1.mel_cepstrum.data, radius=1, dist=2)
    d2, _ = ko.fastdtw(exp2, act2.mel_cepstrum.data, radius=1, dist=2)
    assert round_equal(252, d1), d1
    assert round_equal(302, d2), d2


def _run_cli(main, argv):
    old = sys.argv
    sys.argv = ['prog'] + argv
    try:
        main()
    finally:
        sys.argv = old


def test_cli_resynthesis_kat(kwiiyatta, tmp_path):
    """test_resynthesize_voice.py:19-39 and :42-62"""
    import kwiiyatta_amd.resynthesize_voice as rv
    _run_cli(rv.main, ['--result-dir', str(tmp_path), CLB_WAV])
    out = tmp_path / 'arctic_a0001.wav'
    assert out.is_file()
    f0d, spd, apd, mcd = feature_diffs(kwiiyatta.analyze_wav(CLB_WAV), kwiiyatta.analyze_wav(out))
    assert round_equal(0.079, f0d) and round_equal(0.20, spd)
    assert round_equal(0.073, apd) and round_equal(0.054, mcd)

    _run_cli(rv.main, ['--result-dir', str(tmp_path), '--mcep', CLB_WAV])
    f0d, spd, apd, mcd = feature_diffs(kwiiyatta.analyze_wav(CLB_WAV), kwiiyatta.analyze_wav(out))
    assert round_equal(0.081, f0d) and round_equal(0.22, spd)
    assert round_equal(0.087, apd) and round_equal(0.051, mcd)


def test_cli_resynthesis_carrier_kat(kwiiyatta, tmp_path):
    """test_resynthesize_voice.py:65-91: 0.093 / 0.23 / 0.093 / 0.060"""
    import kwiiyatta_amd.resynthesize_voice as rv
    _run_cli(rv.main, [CLB_WAV, '--result-dir', str(tmp_path), '--mcep', '--mcep-order', '48',
                       '--carrier', SLT_WAV])
    clb, slt = kwiiyatta.analyze_wav(CLB_WAV), kwiiyatta.analyze_wav(SLT_WAV)
    expected = kwiiyatta.align(clb, slt)
    expected.f0 = slt.f0
    actual = kwiiyatta.analyze_wav(tmp_path / 'arctic_a0001.wav')
    f0d, spd, apd, mcd = feature_diffs(expected, actual)
    # the silence padding draws from numpy's unseeded global RNG in the reference
    # test as well; the published digits are stable under it
    assert round_equal(0.093, f0d), f0d
    assert round_equal(0.23, spd), spd
    assert round_equal(0.093, apd), apd
    assert round_equal(0.060, mcd), mcd


def test_cli_resynthesis_diffvc_kat(kwiiyatta, tmp_path):
    """test_resynthesize_voice.py:92-124 (MLSA differential filter on the carrier waveform):
    0.10 / 0.36 / 0.076 / 0.081"""
    import kwiiyatta_amd.resynthesize_voice as rv
    # the silence padding of align() draws from numpy's global RNG (unseeded in the reference test);
    # the mel-cepstrum figure moves between 0.0809 and 0.0812 with it, so the draw is fixed here
    np.random.seed(0)
    _run_cli(rv.main, [CLB_WAV, '--result-dir', str(tmp_path), '--mcep', '--mcep-order', '48',
                       '--carrier', SLT_WAV, '--diffvc'])
    clb, slt = kwiiyatta.analyze_wav(CLB_WAV), kwiiyatta.analyze_wav(SLT_WAV)
    expected = kwiiyatta.align(clb, slt)
    expected.f0 = slt.f0
    sp = np.array(expected.spectrum_envelope)
    sp *= np.exp(np.mean(np.log(slt.spectrum_envelope), axis=1) - np.mean(np.log(sp), axis=1)).reshape(-1, 1)
    expected.spectrum_envelope = sp
    expected.aperiodicity = slt.aperiodicity
    expected.mel_cepstrum = None
    actual = kwiiyatta.analyze_wav(tmp_path / 'arctic_a0001.wav')
    f0d, spd, apd, mcd = feature_diffs(expected, actual)
    assert round_equal(0.10, f0d), f0d
    assert round_equal(0.36, spd), spd
    assert round_equal(0.076, apd), apd
    assert round_equal(0.081, mcd), mcd


def test_cli_voice_conversion_kat(kwiiyatta, tmp_path):
    """test_convert_voice.py:76-129 (16 kHz set-up): train on 8 pairs with one
    component and seed 0, convert a0009; .synth.wav envelope
    f0 [0.10,0.12] spec [0.47,0.52] ap [0.073,0.095] mcep [0.078,0.11], and the .diff.wav envelope."""
    import kwiiyatta_amd.convert_voice as cv
    src = tmp_path / 'src'
    src.mkdir()
    for n in range(1, 9):
        shutil.copy(pathlib.Path(CLB_DIR) / f'arctic_a{n:04}.wav', src)
    res = tmp_path / 'result'
    np.random.seed(0)
    _run_cli(cv.main, ['--source', str(src), '--target', SLT_DIR, '--result-dir', str(res),
                       '--converter-seed', '0', '--converter-components', '1', '--max-files', '8',
                       str(pathlib.Path(CLB_DIR) / 'arctic_a0009.wav')])
    out = res / 'arctic_a0009.synth.wav'
    assert out.is_file() and (res / 'arctic_a0009.diff.wav').is_file()
    # expected feature (test_convert_voice.py:26-38): the source analysis with the
    # DTW-aligned target spectrum, re-scaled frame by frame to the source power
    clb = kwiiyatta.analyze_wav(pathlib.Path(CLB_DIR) / 'arctic_a0009.wav')
    slt = kwiiyatta.analyze_wav(pathlib.Path(SLT_DIR) / 'arctic_a0009.wav')
    tgt_aligned = kwiiyatta.align(slt, clb)
    expected = kwiiyatta.feature(clb)
    sp = np.array(tgt_aligned.spectrum_envelope)
    sp *= np.exp(np.mean(np.log(clb.spectrum_envelope), axis=1)
                 - np.mean(np.log(sp), axis=1)).reshape(-1, 1)
    expected.spectrum_envelope = sp
    act = kwiiyatta.analyze_wav(out)
    f0d, spd, apd, mcd = feature_diffs(expected, act)
    assert 0.10 < f0d < 0.12, f0d
    assert 0.47 < spd < 0.52, spd
    assert 0.073 < apd < 0.095, apd
    assert 0.078 < mcd < 0.11, mcd
    # the differential (MLSA-filtered) output: f0 [0.057,0.092] spec [0.46,0.54] ap [0.042,0.049] mcep [0.077,0.11]
    f0d, spd, apd, mcd = feature_diffs(expected, kwiiyatta.analyze_wav(res / 'arctic_a0009.diff.wav'))
    assert 0.057 < f0d < 0.092, f0d
    assert 0.46 < spd < 0.54, spd
    assert 0.042 < apd < 0.049, apd
    assert 0.077 < mcd < 0.11, mcd


def test_cli_converter_model_and_batch(kwiiyatta, request, tmp_path):
    """Additions to the reference's command line: `--converter-model` writes the trained converter and a later run
    loads it instead of training (no --source / --target): same outputs bit for bit.  `--batch` renders the
    .synth.wav outputs through the HBM-resident batch path: same samples as file by file."""
    import kwiiyatta_amd.convert_voice as cv
    from scipy.io import wavfile as sio
    src = tmp_path / 'src'
    src.mkdir()
    for n in range(1, 5):
        shutil.copy(pathlib.Path(CLB_DIR) / f'arctic_a{n:04}.wav', src)
    model = tmp_path / 'converter.npz'
    inputs = [str(pathlib.Path(CLB_DIR) / f'arctic_a{n:04}.wav') for n in (8, 9)]
    np.random.seed(0)
    _run_cli(cv.main, ['--source', str(src), '--target', SLT_DIR, '--result-dir', str(tmp_path / 'a'),
                       '--converter-seed', '0', '--converter-components', '2', '--max-files', '4',
                       '--converter-model', str(model)] + inputs)
    assert model.is_file()
    _run_cli(cv.main, ['--result-dir', str(tmp_path / 'b'), '--converter-model', str(model)] + inputs)
    for name in ('arctic_a0008', 'arctic_a0009'):
        for kind in ('synth', 'diff'):
            fa, a = sio.read(tmp_path / 'a' / f'{name}.{kind}.wav')
            fb, b = sio.read(tmp_path / 'b' / f'{name}.{kind}.wav')
            assert fa == fb and np.array_equal(a, b), (name, kind)
    if request.node.callspec.params['kwiiyatta'] == 'hip':
        # the batch path: wav in -> 16-bit PCM out on the device, both outputs of every file
        _run_cli(cv.main, ['--result-dir', str(tmp_path / 'c'), '--converter-model', str(model), '--batch'] + inputs)
        for name in ('arctic_a0008', 'arctic_a0009'):
            for kind in ('synth', 'diff'):
                _, a = sio.read(tmp_path / 'a' / f'{name}.{kind}.wav')
                _, c = sio.read(tmp_path / 'c' / f'{name}.{kind}.wav')
                assert a.shape == c.shape and c.dtype == np.int16
                assert np.abs(a.astype(np.int64) - c.astype(np.int64)).max() <= 1, (name, kind)      # 16-bit samples
        _run_cli(cv.main, ['--result-dir', str(tmp_path / 'd'), '--converter-model', str(model), '--batch',
                           '--no-diffvc'] + inputs)
        for name in ('arctic_a0008', 'arctic_a0009'):
            _, c = sio.read(tmp_path / 'c' / f'{name}.synth.wav')
            _, d = sio.read(tmp_path / 'd' / f'{name}.synth.wav')
            assert np.array_equal(c, d)
        assert not (tmp_path / 'd' / 'arctic_a0009.diff.wav').exists()


def test_converter_stack_checks(kwiiyatta):
    """test_converter.py:28-62: error messages and identity pass-through"""
    import kwiiyatta_amd.converter.abc as cabc
    from kwiiyatta_amd.converter import (DeltaFeatureConverter, MelCepstrumDataset,
                                         MelCepstrumFeatureConverter)

    class Nop(cabc.FeatureConverter):
        def _train(self, dataarray):
            pass

        def convert(self, feature):
            return feature

    a = kwiiyatta.analyze_wav(CLB_WAV)
    dc = DeltaFeatureConverter(Nop())
    dc.train(MelCepstrumDataset({'key': a}), ['key'])
    a3 = kwiiyatta.analyze_wav(CLB_WAV, frame_period=3)
    with pytest.raises(ValueError) as e:
        dc.convert(a3.mel_cepstrum.data[:, 1:], a3)
    assert str(e.value) == 'frame_period is expected to 5 but 3'
    mc = a.mel_cepstrum.data[:, 1:]
    assert (dc.convert(mc, a) == mc).all()

    f = kwiiyatta.feature(a)
    mcc = MelCepstrumFeatureConverter(Nop())
    mcc.train({'key': copy.copy(f)}, ['key'])
    f.mel_cepstrum_order = 32
    with pytest.raises(ValueError) as e:
        mcc.convert(f.mel_cepstrum)
    assert str(e.value) == 'order is expected to 24 but 32'
    f.mel_cepstrum_order = 24
    assert (mcc.convert(f.mel_cepstrum).data == f.mel_cepstrum.data).all()


def test_datasets(kwiiyatta, tmp_path):
    """test_dataset.py: key sets, lazy loading, parallel/aligned stacking"""
    from kwiiyatta_amd.converter import DeltaFeatureDataset, MelCepstrumDataset, make_dataset_to_array
    src = tmp_path / 'src'
    src.mkdir()
    for n in (1, 2, 3):
        shutil.copy(pathlib.Path(CLB_DIR) / f'arctic_a{n:04}.wav', src)
    ds = kwiiyatta.WavFileDataset(src)
    assert {str(k) for k in ds.keys()} == {f'arctic_a{n:04}.wav' for n in (1, 2, 3)}
    with pytest.raises(FileNotFoundError):
        kwiiyatta.WavFileDataset(tmp_path / 'missing')
    with pytest.raises(NotADirectoryError):
        kwiiyatta.WavFileDataset(src / 'arctic_a0001.wav')
    par = kwiiyatta.ParallelDataset(ds, kwiiyatta.WavFileDataset(pathlib.Path(SLT_DIR)))
    assert len(par) == 3
    np.random.seed(0)
    aligned = kwiiyatta.align(ds, kwiiyatta.WavFileDataset(pathlib.Path(SLT_DIR)))
    key = sorted(aligned.keys())[0]
    a, b = aligned[key]
    assert a.frame_len == b.frame_len > 100
    arr = make_dataset_to_array(DeltaFeatureDataset(MelCepstrumDataset(aligned)), [key])
    assert arr.shape[1] == 2 * 3 * 24 and arr.shape[0] <= a.frame_len
