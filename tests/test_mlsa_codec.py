"""MLSA differential filter and the aperiodicity band codec (8f-3).

CPU: properties that pin the oracle restatement independently of upstream vectors (the
reference's own MLSA known-answer tests run in tests/test_package.py).  GPU: the HIP kernels
through the C ABI against the oracle.
"""
import numpy as np
import pytest
from scipy.io import wavfile

from conftest import CLB_WAV, clb_variant


def _mcep(rng, order, scale=0.3):
    mc = np.zeros(order + 1)
    mc[1:] = rng.standard_normal(order) * scale / np.arange(1, order + 1)
    return mc


# ----------------------------------------------------------------------------------- CPU
def test_oracle_mlsa_filter_realises_the_mel_cepstral_envelope():
    """White noise through the MLSA filter of a constant mel-cepstrum acquires the power spectrum
    exp(2 Re F) = mc2sp(mc) (Pade order 4: within a few hundredths of a dB on average)."""
    import scipy.signal as ss
    from oracle import oracle as ko
    fs, order, hop, T = 16000, 24, 80, 400
    alpha = ko.mcepalpha(fs)
    rng = np.random.default_rng(0)
    mc = _mcep(rng, order)
    x = rng.standard_normal(T * hop + 10)
    y = ko.mlsa_synthesis(x, ko.mc2b(np.tile(mc, (T, 1)), alpha), alpha, hop)
    _, pyy = ss.welch(y[2000:], fs, nperseg=1024)
    _, pxx = ss.welch(x[2000:], fs, nperseg=1024)
    h = ko.mc2sp(mc[None, :], alpha, 1024)[0]
    err_db = np.abs(10 * np.log10(pyy / pxx) - 10 * np.log10(h))
    assert err_db.mean() <= 0.05 and err_db.max() <= 0.5, (err_db.mean(), err_db.max())
    # frames that reach the end of the signal are not processed
    assert np.all(y[T * hop:] == 0.0)


def test_oracle_mc2b_inverts_b2mc():
    from oracle import oracle as ko
    rng = np.random.default_rng(1)
    mc = rng.standard_normal((5, 25))
    b = ko.mc2b(mc, 0.41)
    back = b.copy()
    back[:, :-1] += 0.41 * b[:, 1:]          # SPTK b2mc
    assert np.abs(back - mc).max() <= 1e-14


def test_oracle_codec_round_trip_and_vuv():
    from oracle import oracle as ko
    fs, fft = 48000, 2048
    K = fft // 2 + 1
    nb = 5
    rng = np.random.default_rng(2)
    coded = -rng.uniform(3.0, 40.0, (6, nb))
    coded[2] = -0.1                                   # mean above -0.5 dB: decoded as unvoiced
    ap = ko.decode_aperiodicity(coded, fs, fft)
    assert ap.shape == (6, K)
    assert np.all(ap[2] == 1 - 1e-12)
    again = ko.code_aperiodicity(ap, fs)
    assert again.shape == (6, nb)
    keep = [0, 1, 3, 4, 5]
    assert np.abs(again[keep] - coded[keep]).max() <= 1e-9   # the band centres fall on bins: exact round trip
    assert np.abs(20 * np.log10(ap[keep, 0]) + 60.0).max() <= 1e-9


# ----------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize('fs,order,pd', [(16000, 24, 4), (48000, 48, 4), (16000, 24, 5), (16000, 2, 4), (16000, 9, 5),
                                         (16000, 10, 4), (22050, 33, 5), (44100, 41, 4), (48000, 57, 5),
                                         (48000, 62, 4)])
def test_hip_mlsa_matches_oracle(fs, order, pd):
    from oracle import oracle as ko
    from kwiiyatta_amd.backend import sptk
    rng = np.random.default_rng(3)
    hop = fs // 200
    T = 120
    alpha = ko.mcepalpha(fs)
    mc = np.stack([_mcep(rng, order, 0.4) for _ in range(T)])
    mc[:, 0] = 0.0
    b = sptk.mc2b(mc, alpha)
    assert np.abs(b - ko.mc2b(mc, alpha)).max() <= 1e-15
    x = rng.standard_normal(T * hop - 37) * 0.1          # the last frame reaches the end: unprocessed
    got = sptk.Synthesizer(sptk.MLSADF(order=order, alpha=alpha, pd=pd), hopsize=hop).synthesis(x, b)
    ref = ko.mlsa_synthesis(x, b, alpha, hop, pd)
    assert got.shape == ref.shape
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 1e-11 * scale, np.abs(got - ref).max() / scale
    assert np.all(got[(T - 1) * hop:] == 0.0)


@pytest.mark.gpu
def test_hip_mlsa_on_speech():
    from oracle import oracle as ko
    from kwiiyatta_amd.backend import sptk
    fs, d = wavfile.read(CLB_WAV)
    x = np.ascontiguousarray(d.astype(np.float64) / 2 ** 15)
    rng = np.random.default_rng(4)
    hop, order = 80, 24
    T = len(x) // hop + 1
    alpha = ko.mcepalpha(fs)
    walk = np.cumsum(rng.standard_normal((T, order)) * 0.02, axis=0) / np.arange(1, order + 1)
    b = ko.mc2b(np.hstack([np.zeros((T, 1)), walk]), alpha)
    got = sptk.Synthesizer(sptk.MLSADF(order, alpha), hop).synthesis(x, b)
    ref = ko.mlsa_synthesis(x, b, alpha, hop)
    assert np.sqrt(np.mean((got - ref) ** 2)) <= 1e-12 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize('path', [CLB_WAV, clb_variant('48')])
def test_hip_codec_matches_oracle(path):
    from oracle import oracle as ko
    from kwiiyatta_amd.backend import world as kw
    fs, d = wavfile.read(path)
    x = np.ascontiguousarray(d[:fs].astype(np.float64) / 2 ** 15)
    f0, t = ko.dio(x, fs)
    ap = ko.d4c(x, ko.stonemask(x, f0, t, fs), t, fs)
    coded, ref = kw.code_aperiodicity(ap, fs), ko.code_aperiodicity(ap, fs)
    assert coded.shape == ref.shape == (len(f0), kw.get_num_aperiodicities(fs))
    assert np.abs(coded - ref).max() <= 1e-9
    for fft in ((ap.shape[1] - 1) * 2, 1024):
        got, want = kw.decode_aperiodicity(ref, fs, fft), ko.decode_aperiodicity(ref, fs, fft)
        assert got.shape == want.shape
        assert np.array_equal(got == 1 - 1e-12, want == 1 - 1e-12)     # same frames kept unvoiced
        assert np.abs(got - want).max() <= 1e-12
    # a truncated band set, as the reference passes when it lowers the sampling rate (world.py:121-128)
    if coded.shape[1] > 1:
        got = kw.decode_aperiodicity(np.ascontiguousarray(ref[:, :1]), 16000, 1024)
        assert np.abs(got - ko.decode_aperiodicity(np.ascontiguousarray(ref[:, :1]), 16000, 1024)).max() <= 1e-12
