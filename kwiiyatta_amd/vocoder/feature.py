"""Concrete, mutable feature container plus the `feature()` / `pad_silence()`
helpers (mirrors /root/reference/kwiiyatta/vocoder/feature.py:10-91)."""
import copy

import numpy as np

import kwiiyatta_amd as kwiiyatta
from . import abc


def feature(arg, **kwargs):
    """feature(fs) -> empty Feature; feature(other_feature) -> materialised copy."""
    if isinstance(arg, int):
        return Feature(arg, **kwargs)
    if isinstance(arg, abc.Feature):
        return Feature.init(arg, **kwargs)
    raise TypeError("argument should be int or Feature")


def pad_silence(feature, frame_len):
    """`frame_len` frames of silence before and after the feature."""
    syn, fs, n_bins = feature.Synthesizer, feature.fs, feature.spectrum_len
    padded = kwiiyatta.feature(feature)
    # draw order matters for reproducibility under a seeded numpy RNG:
    # leading block first, then trailing block, spectrum only
    padded.f0 = np.concatenate((syn.silence_f0(frame_len, fs), feature.f0,
                                syn.silence_f0(frame_len, fs)))
    head = syn.silence_spectrum_envelope(frame_len, fs, n_bins)
    body = feature.spectrum_envelope
    tail = syn.silence_spectrum_envelope(frame_len, fs, n_bins)
    padded.spectrum_envelope = np.concatenate((head, body, tail))
    padded.aperiodicity = np.concatenate((syn.silence_aperiodicity(frame_len, fs, n_bins),
                                          feature.aperiodicity,
                                          syn.silence_aperiodicity(frame_len, fs, n_bins)))
    return padded


class Feature(abc.MutableFeature):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._f0 = self._spectrum_envelope = self._aperiodicity = None

    @classmethod
    def init(cls, feature, **kwargs):
        """Copy another feature's arrays (forces their extraction)."""
        kwargs.setdefault('frame_period', feature.frame_period)
        kwargs.setdefault('mcep_order', feature.mel_cepstrum_order)
        kwargs.setdefault('Synthesizer', feature.Synthesizer)
        other = cls(feature.fs, **kwargs)
        other._f0 = feature.f0
        other._spectrum_envelope = feature.spectrum_envelope
        other._aperiodicity = feature.aperiodicity
        other._mel_cepstrum = copy.copy(feature._mel_cepstrum)
        return other

    @property
    def spectrum_len(self):
        for arr in (self._spectrum_envelope, self._aperiodicity):
            if arr is not None:
                return arr.shape[-1]
        return super().spectrum_len

    def _get_f0(self):
        return self._f0

    def _set_f0(self, value):
        self._f0 = value

    def _get_spectrum_envelope(self):
        return self._spectrum_envelope

    def _set_spectrum_envelope(self, value):
        self._spectrum_envelope = value

    def _get_aperiodicity(self):
        return self._aperiodicity

    def _set_aperiodicity(self, value):
        self._aperiodicity = value

    def synthesize(self):
        return self.Synthesizer.synthesize(self)
