"""The materialised feature set -- all slots assignable, nothing computed behind the caller's back except the
spectrum / mel-cepstrum conversions -- with its two constructors (`feature(fs)` empty, `feature(other)` a
snapshot) and `pad_silence`.  API of kwiiyatta.vocoder.feature (/root/reference/kwiiyatta/vocoder/feature.py)."""
import copy

import numpy as np

from . import abc


class Feature(abc.MutableFeature):
    @classmethod
    def init(cls, feature, **kwargs):
        """Snapshot of another feature set: its arrays (extracted now if it is an analyzer) are shared, not copied;
        frame period, mel-cepstrum order and vocoder are inherited unless overridden."""
        inherited = dict(frame_period=feature.frame_period, mcep_order=feature.mel_cepstrum_order,
                         Synthesizer=feature.Synthesizer)
        snap = cls(feature.fs, **{**inherited, **kwargs})
        for slot in abc.ARRAY_SLOTS:
            snap._put(slot, getattr(feature, slot))
        snap._mel_cepstrum = copy.copy(feature._mel_cepstrum)
        return snap

    def synthesize(self):
        return self.Synthesizer.synthesize(self)


def feature(arg, **kwargs):
    """feature(48000) -> an empty feature set at that sampling rate; feature(f) -> snapshot of f"""
    if isinstance(arg, int):
        return Feature(arg, **kwargs)
    if isinstance(arg, abc.Feature):
        return Feature.init(arg, **kwargs)
    raise TypeError("argument should be int or Feature")


def pad_silence(feature, frame_len):
    """`frame_len` frames of the vocoder's silence before and after the feature set.  The silent envelopes are
    random (numpy's global generator): leading block first, then the trailing one -- the order the reference
    draws them in, which seeded runs depend on."""
    vocoder, fs, bins = feature.Synthesizer, feature.fs, feature.spectrum_len

    def framed(quiet, body):
        head = quiet()
        return np.concatenate((head, body, quiet()))

    padded = Feature.init(feature)
    padded.f0 = framed(lambda: vocoder.silence_f0(frame_len, fs), feature.f0)
    padded.spectrum_envelope = framed(lambda: vocoder.silence_spectrum_envelope(frame_len, fs, bins),
                                      feature.spectrum_envelope)
    padded.aperiodicity = framed(lambda: vocoder.silence_aperiodicity(frame_len, fs, bins), feature.aperiodicity)
    return padded
