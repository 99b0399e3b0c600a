"""Vocoder layer: feature sets, the WORLD vocoder that computes and renders them, alignment.
Export list of kwiiyatta.vocoder (/root/reference/kwiiyatta/vocoder/__init__.py)."""
import copy

from . import abc
from .align import align, align_even
from .feature import Feature, feature, pad_silence
from .mcep import MelCepstrum
from .world import WorldAnalyzer, WorldSynthesizer

Analyzer = WorldAnalyzer            # the vocoder in use; analyze_wav() reads it at call time
Synthesizer = WorldSynthesizer


def analyze_wav(wavfile, Analyzer=None, **kwargs):
    """lazy analysis of a wav file (nothing is computed until a feature is asked for)"""
    chosen = Analyzer if Analyzer is not None else globals()['Analyzer']
    return chosen.load_wav(wavfile, **kwargs)


def resample(feature, new_fs, **kwargs):
    """a copy of a feature set or of a MelCepstrum at another sampling rate"""
    if isinstance(feature, abc.Feature):
        result = Feature.init(feature)
    elif isinstance(feature, MelCepstrum):
        result = copy.copy(feature)
    else:
        raise TypeError("argument should be Feature or MelCepstrum")
    if feature.fs != new_fs:
        result.resample(new_fs, **kwargs)
    return result


def reshape(feature, reshape_spectrum_len):
    """a copy of a feature set with another number of spectral bins"""
    result = Feature.init(feature)
    if feature.spectrum_len != reshape_spectrum_len:
        result.reshape(reshape_spectrum_len)
    return result


__all__ = ['analyze_wav', 'resample', 'reshape', 'align', 'align_even', 'Feature', 'feature', 'pad_silence',
           'MelCepstrum', 'Analyzer', 'Synthesizer']
