"""Differential-spectrum filtering: a waveform is passed through the MLSA filter described by a mel-cepstrum (for
voice conversion: the difference between target and source mel-cepstra), which changes its spectral envelope
without re-synthesising it.  API of kwiiyatta.filter (/root/reference/kwiiyatta/filter/mlsa.py:9-30).  mc2b and the
filter itself are HIP kernels behind the pysptk-shaped classes of kwiiyatta_amd.backend.sptk."""
import numpy as np

from ..backend import sptk


def _at_waveform_rate(mcep, fs):
    """the mel-cepstrum re-expressed at the waveform's sampling rate"""
    import kwiiyatta_amd as k
    if mcep.fs > fs:
        return k.resample(mcep, fs)
    if mcep.fs == fs:
        return mcep
    # lower rate: its spectrum covers only the lower band of the waveform's; the band above is held at the level of
    # the last bin the mel-cepstrum knows about (a flat continuation of the filter's gain)
    wide = k.Synthesizer.resample_spectrum_envelope(mcep.extract_spectrum(), mcep.fs, fs)
    known = mcep.fs * wide.shape[1] // fs
    wide[:, known:] = wide[:, known - 1:known]
    out = k.MelCepstrum(fs, mcep.frame_period)
    out.extract(wide)
    return out


def apply_mlsa_filter(wav, mcep):
    """`wav` (Wavdata or anything with .fs / .data) filtered frame by frame with `mcep` (c0 ignored)"""
    import kwiiyatta_amd as k
    mcep = _at_waveform_rate(mcep, wav.fs)
    shape_only = np.array(mcep.data, dtype=np.float64)
    shape_only[:, 0] = 0.0                               # the filter changes the envelope's shape, not its power
    alpha = mcep.alpha()
    hop = int(mcep.fs * (mcep.frame_period * 0.001))
    engine = sptk.Synthesizer(sptk.MLSADF(order=mcep.order, alpha=alpha), hopsize=hop)
    return k.Wavdata(wav.fs, engine.synthesis(wav.data, sptk.mc2b(shape_only, alpha=alpha)))


__all__ = ['apply_mlsa_filter']
