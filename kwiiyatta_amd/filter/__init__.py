"""Differential MLSA filtering: kwiiyatta.filter of the reference
(/root/reference/kwiiyatta/filter/mlsa.py:9-30), SURVEY.md section 8(f)-2.

Same flow as the reference: bring the mel-cepstrum to the waveform's sampling rate, zero the
power coefficient, mc2b, and run the waveform through the MLSA filter whose coefficients are
interpolated inside each frame.  mc2b and the filter are the HIP kernels behind the
pysptk-shaped shims of kwiiyatta_amd.backend.sptk."""
import numpy as np

from ..backend import sptk as pysptk
from ..backend.sptk import MLSADF, Synthesizer


def apply_mlsa_filter(wav, mcep):
    import kwiiyatta_amd as kwiiyatta
    if mcep.fs > wav.fs:
        mcep = kwiiyatta.resample(mcep, wav.fs)
    elif mcep.fs < wav.fs:
        spec = kwiiyatta.Synthesizer.resample_spectrum_envelope(
            mcep.extract_spectrum(),
            mcep.fs,
            wav.fs
        )
        cutoff = mcep.fs * spec.shape[1] // wav.fs
        spec[:, cutoff:] = np.tile(np.atleast_2d(spec[:, cutoff - 1]).T, spec.shape[-1] - cutoff)
        mcep = kwiiyatta.MelCepstrum(wav.fs, mcep.frame_period)
        mcep.extract(spec)
    # remove power coefficients
    mc = np.hstack((np.zeros((mcep.data.shape[0], 1)), mcep.data[:, 1:]))
    alpha = mcep.alpha()
    engine = Synthesizer(MLSADF(order=mcep.order, alpha=alpha),
                         hopsize=int(mcep.fs * (mcep.frame_period * 0.001)))
    b = pysptk.mc2b(mc.astype(np.float64), alpha=alpha)
    waveform = engine.synthesis(wav.data, b)
    return kwiiyatta.Wavdata(wav.fs, waveform)


__all__ = ['apply_mlsa_filter']
