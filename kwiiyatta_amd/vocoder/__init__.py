"""Vocoder layer: feature model, WORLD backend, mel-cepstrum, DTW alignment
(mirrors the export list of /root/reference/kwiiyatta/vocoder/__init__.py)."""
import copy

from .align import align, align_even
from .feature import Feature, feature, pad_silence
from .mcep import MelCepstrum
from .world import WorldAnalyzer, WorldSynthesizer
from . import abc

Analyzer = WorldAnalyzer
Synthesizer = WorldSynthesizer


def analyze_wav(wavfile, Analyzer=None, **kwargs):
    cls = Analyzer if Analyzer is not None else globals()['Analyzer']
    return cls.load_wav(wavfile, **kwargs)


def resample(feature, new_fs, **kwargs):
    if isinstance(feature, abc.Feature):
        f = Feature.init(feature)
        if feature.fs != new_fs:
            f.resample(new_fs)
        return f
    if isinstance(feature, MelCepstrum):
        m = copy.copy(feature)
        if feature.fs != new_fs:
            m.resample(new_fs, **kwargs)
        return m
    raise TypeError("argument should be Feature or MelCepstrum")


def reshape(feature, reshape_spectrum_len):
    f = Feature.init(feature)
    if feature.spectrum_len != reshape_spectrum_len:
        f.reshape(reshape_spectrum_len)
    return f


__all__ = ['analyze_wav', 'resample', 'reshape', 'align', 'align_even', 'Feature', 'feature',
           'pad_silence', 'MelCepstrum', 'Analyzer', 'Synthesizer']
