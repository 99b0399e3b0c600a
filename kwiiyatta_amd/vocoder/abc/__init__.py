"""Abstract feature model of the vocoder layer.

Same public surface and semantics as the reference's kwiiyatta.vocoder.abc
package (/root/reference/kwiiyatta/vocoder/abc/{feature,analyzer,synthesizer}.py):

  Feature         lazy, read-only view of (f0, spectrum_envelope, aperiodicity,
                  mel_cepstrum, is_voiced) with caching; slicing / row gather
  MutableFeature  adds setters that keep the caches coherent, resample/reshape
  Analyzer        a Feature computed on demand from a waveform
  Synthesizer     feature -> waveform, plus the spectral-axis reshape/resample
                  helpers and silence generators

Everything here is host-side bookkeeping; the arithmetic is delegated to the
backend modules (GPU).
"""
import abc
import copy

import numpy as np
import scipy.signal

import kwiiyatta_amd as kwiiyatta
from ..mcep import MelCepstrum

__all__ = ['Analyzer', 'Feature', 'MutableFeature', 'Synthesizer']


class Feature(abc.ABC):
    def __init__(self, fs, frame_period=5, mcep_order=24, Synthesizer=None):
        self.mel_cepstrum_order = mcep_order
        self.Synthesizer = kwiiyatta.Synthesizer if Synthesizer is None else Synthesizer
        self._mel_cepstrum = MelCepstrum(fs, frame_period)
        self._is_voiced = None

    def __copy__(self):
        clone = self.__class__.__new__(self.__class__)
        clone.__dict__.update(self.__dict__)
        clone._mel_cepstrum = copy.copy(self._mel_cepstrum)
        return clone

    # ---- basic properties ------------------------------------------------------
    @property
    def fs(self):
        return self._mel_cepstrum.fs

    @property
    def frame_period(self):
        return self._mel_cepstrum.frame_period

    @property
    def frame_len(self):
        return self.f0.shape[0]

    @property
    @abc.abstractmethod
    def spectrum_len(self):
        return self.Synthesizer.fs_spectrum_len(self.fs)

    # ---- storage hooks of the concrete classes ----------------------------------
    @abc.abstractmethod
    def _get_f0(self):
        raise NotImplementedError

    @abc.abstractmethod
    def _get_spectrum_envelope(self):
        raise NotImplementedError

    @abc.abstractmethod
    def _get_aperiodicity(self):
        raise NotImplementedError

    @abc.abstractmethod
    def ascontiguousarray(self):
        raise NotImplementedError

    # ---- features ----------------------------------------------------------------
    @property
    def f0(self):
        return self._get_f0()

    @property
    def spectrum_envelope(self):
        spec = self._get_spectrum_envelope()
        return spec if spec is not None else self.extract_spectrum_envelope()

    @property
    def aperiodicity(self):
        return self._get_aperiodicity()

    def extract_spectrum_envelope(self, spectrum_len=None):
        """Spectrum rebuilt from the stored mel-cepstrum (None if there is none)."""
        if self._mel_cepstrum.data is None:
            return None
        if spectrum_len is None:
            spectrum_len = self.spectrum_len
        return self._mel_cepstrum.extract_spectrum(spectrum_len)

    def extract_mel_cepstrum(self, spectrum=None):
        if spectrum is not None:
            self._set_spectrum_envelope(None)
            source = spectrum
        else:
            have = self._mel_cepstrum
            if have.data is not None:
                if self.mel_cepstrum_order == have.order:
                    return have
                if self.mel_cepstrum_order < have.order:   # a lower order is a prefix
                    cut = copy.copy(have)
                    cut.data = cut.data[:, :self.mel_cepstrum_order + 1]
                    return cut
            source = self.spectrum_envelope
        if source is None:
            return None
        self._mel_cepstrum.extract(source, self.mel_cepstrum_order)
        return self._mel_cepstrum

    def clear_mel_cepstrum(self):
        self._mel_cepstrum.data = None

    @property
    def mel_cepstrum(self):
        return self.extract_mel_cepstrum()

    def extract_is_voiced(self):
        if self._is_voiced is None:
            self._is_voiced = self.Synthesizer.extract_is_voiced(self)
        return self._is_voiced

    @property
    def is_voiced(self):
        return self.extract_is_voiced()

    # ---- spectral-axis conversions -------------------------------------------------
    def reshaped_spectrum_envelope(self, new_spectrum_len):
        spec = self._get_spectrum_envelope()
        if spec is None:
            return self._mel_cepstrum.extract_spectrum(new_spectrum_len)
        return self.Synthesizer.reshape_spectrum_envelope(spec, self.fs, new_spectrum_len)

    def reshaped_aperiodicity(self, new_spectrum_len):
        return self.Synthesizer.reshape_aperiodicity(self.aperiodicity, self.fs, new_spectrum_len)

    def resample_spectrum_envelope(self, new_fs):
        return self.Synthesizer.resample_spectrum_envelope(self.spectrum_envelope, self.fs, new_fs)

    def resample_aperiodicity(self, new_fs):
        return self.Synthesizer.resample_aperiodicity(self.aperiodicity, self.fs, new_fs)

    def resample_mel_cepstrum(self, new_fs):
        if self._get_spectrum_envelope() is not None:
            return kwiiyatta.resample(self.mel_cepstrum, new_fs, order=self.mel_cepstrum_order,
                                      spectrum=self.resample_spectrum_envelope(new_fs))
        return kwiiyatta.resample(self.mel_cepstrum, new_fs, order=self.mel_cepstrum_order,
                                  Synthesizer=self.Synthesizer)

    def synthesize(self, **kwargs):
        return self.Synthesizer.synthesize(self, **kwargs)

    # ---- comparison / indexing ---------------------------------------------------------
    def __eq__(self, other):
        if self.frame_period != other.frame_period or self.fs != other.fs:
            return False
        pairs = ((self._get_f0(), other._get_f0()),
                 (self._get_spectrum_envelope(), other._get_spectrum_envelope()),
                 (self._get_aperiodicity(), other._get_aperiodicity()))
        return not any((a != b).any() for a, b in pairs)

    __hash__ = None

    def __getitem__(self, key):
        f0 = self.f0[key]
        spec = self.spectrum_envelope[key]
        ape = self.aperiodicity[key]
        voiced = self._is_voiced[key] if self._is_voiced is not None else None
        mcep = self.mel_cepstrum.data[key] if self._mel_cepstrum.data is not None else None

        if isinstance(key, int):           # one frame: plain tuple
            if mcep is None:
                mcep = self.mel_cepstrum.data[key]
            return f0, spec, ape, mcep

        picked = kwiiyatta.feature(self)
        picked._f0, picked._spectrum_envelope, picked._aperiodicity = f0, spec, ape
        if voiced is not None:
            picked._is_voiced = voiced
        if mcep is not None:
            picked._mel_cepstrum.data = mcep
        return picked


class MutableFeature(Feature):
    @abc.abstractmethod
    def _set_f0(self, value):
        raise NotImplementedError

    @abc.abstractmethod
    def _set_spectrum_envelope(self, value):
        raise NotImplementedError

    @abc.abstractmethod
    def _set_aperiodicity(self, value):
        raise NotImplementedError

    @Feature.f0.setter
    def f0(self, value):
        if value is not None:
            self._is_voiced = None
        self._set_f0(value)

    @Feature.spectrum_envelope.setter
    def spectrum_envelope(self, value):
        if value is None:
            # dropping the spectrum: keep its information as a mel-cepstrum
            if self._mel_cepstrum.data is None:
                self.extract_mel_cepstrum()
        else:
            self._mel_cepstrum.data = None
            self._is_voiced = None
        self._set_spectrum_envelope(value)

    @Feature.aperiodicity.setter
    def aperiodicity(self, value):
        if value is not None:
            self._is_voiced = None
        self._set_aperiodicity(value)

    @Feature.mel_cepstrum.setter
    def mel_cepstrum(self, data):
        if data is not None:
            if isinstance(data, MelCepstrum):
                if data.fs != self.fs:
                    data = data.resample_data(self.fs, Synthesizer=self.Synthesizer)
                else:
                    data = data.data
            elif not isinstance(data, np.ndarray):
                raise TypeError('Feature.mel_cepstrum should be a MelCepstrum or ndarray')
            if data is not None:
                self._set_spectrum_envelope(None)   # the mel-cepstrum is now authoritative
        self._mel_cepstrum.data = data

    def reshape(self, new_spectrum_len):
        if self._get_spectrum_envelope().shape[1] != new_spectrum_len:
            self._set_spectrum_envelope(self.reshaped_spectrum_envelope(new_spectrum_len))
        if self._get_aperiodicity().shape[1] != new_spectrum_len:
            self._set_aperiodicity(self.reshaped_aperiodicity(new_spectrum_len))

    def resample(self, new_fs):
        if new_fs == self.fs:
            return
        # f0 is independent of the sampling rate
        if self._get_aperiodicity() is not None:
            self._set_aperiodicity(self.resample_aperiodicity(new_fs))
        if self._get_spectrum_envelope() is not None:
            self._set_spectrum_envelope(self.resample_spectrum_envelope(new_fs))
            self._mel_cepstrum.data = None
        elif self._mel_cepstrum.data is not None:
            self._mel_cepstrum.resample(new_fs, Synthesizer=self.Synthesizer)
        self._mel_cepstrum._fs = new_fs

    def ascontiguousarray(self):
        self._set_f0(np.ascontiguousarray(self.f0))
        self._set_spectrum_envelope(np.ascontiguousarray(self.spectrum_envelope))
        self._set_aperiodicity(np.ascontiguousarray(self.aperiodicity))


class Analyzer(Feature):
    """A Feature whose arrays are extracted lazily from a waveform."""

    def __init__(self, wavdata, **kwargs):
        super().__init__(wavdata.fs, **kwargs)
        self._data = wavdata.data
        self._f0 = self._spectrum_envelope = self._aperiodicity = self._is_voiced = None

    @classmethod
    def load_wav(cls, wavfile, **kwargs):
        return cls(kwiiyatta.load_wav(wavfile), **kwargs)

    @property
    def data(self):
        return self._data

    @property
    def wavdata(self):
        return kwiiyatta.Wavdata(self.fs, self.data)

    @property
    @abc.abstractmethod
    def frame_len(self):
        raise NotImplementedError

    @abc.abstractmethod
    def clear_features(self):
        self.clear_mel_cepstrum()

    @abc.abstractmethod
    def extract_f0(self):
        raise NotImplementedError

    @abc.abstractmethod
    def extract_spectrum_envelope(self):
        raise NotImplementedError

    @abc.abstractmethod
    def extract_aperiodicity(self):
        raise NotImplementedError

    def _get_f0(self):
        return self.extract_f0()

    def _get_spectrum_envelope(self):
        return self.extract_spectrum_envelope()

    def _get_aperiodicity(self):
        return self.extract_aperiodicity()

    def ascontiguousarray(self):
        self._f0 = np.ascontiguousarray(self.f0)
        self._spectrum_envelope = np.ascontiguousarray(self.spectrum_envelope)
        self._aperiodicity = np.ascontiguousarray(self.aperiodicity)


class Synthesizer(abc.ABC):
    @classmethod
    def synthesize(cls, feature, normalize=True, **kwargs):
        wavdata = cls._synthesize(feature)
        if normalize:
            wavdata.normalize(None)
            fs = feature.fs
            # the reference limits 1-ms chunks, but only frame_len of them (quirk kept)
            for i in range(feature.frame_len):
                kwiiyatta.wavfile.normalize_data(wavdata.data[fs * i // 1000: fs * (i + 1) // 1000],
                                                 **kwargs)
        return wavdata

    @staticmethod
    @abc.abstractmethod
    def _synthesize(feature):
        raise NotImplementedError

    @staticmethod
    def fs_spectrum_len(fs):
        raise NotImplementedError

    # ---- change of the number of spectral bins at a fixed sampling rate -------------------
    @staticmethod
    def _reshape_feature(feature, fs, new_spectrum_len):
        old_len = feature.shape[1]
        g = np.gcd(old_len, new_spectrum_len)
        pad = old_len // g * 20
        trim = new_spectrum_len // g * 20
        first = np.repeat(feature[:, :1], pad, axis=1)
        last = np.repeat(feature[:, -1:], pad, axis=1)
        stretched = scipy.signal.resample_poly(np.hstack((first, feature, last)),
                                               new_spectrum_len, old_len, axis=1)
        return stretched[:, trim:-trim]

    @classmethod
    def reshape_spectrum_envelope(cls, feature, fs, new_spectrum_len):
        return np.exp(cls._reshape_feature(np.log(feature), fs, new_spectrum_len))

    @classmethod
    def reshape_aperiodicity(cls, feature, fs, new_spectrum_len):
        return np.exp(cls._reshape_feature(np.log(feature), fs, new_spectrum_len))

    # ---- change of sampling rate (bins keep their frequency spacing) ------------------------
    @staticmethod
    def _resample_spectrum_len(feature, fs, new_fs):
        return feature.shape[1] * new_fs // fs

    @staticmethod
    def _resample_up(feature, fs, new_fs, new_spectrum_len, pad, window=1):
        extra = new_spectrum_len - feature.shape[1]
        feature = np.hstack((feature, pad[:, -extra:]))
        overlap = pad.shape[1] - extra
        if overlap > 0:
            sl = slice(-extra - overlap, -extra)
            feature[:, sl] *= 1 - window
            feature[:, sl] += window * pad[:, :overlap]
        return feature

    @staticmethod
    def _resample_down(feature, fs, new_fs, new_spectrum_len):
        return feature[:, :new_spectrum_len]

    @staticmethod
    @abc.abstractmethod
    def _resample_up_spectrum_envelope(feature, fs, new_fs, new_spectrum_len):
        raise NotImplementedError

    _resample_down_spectrum_envelope = _resample_down

    @staticmethod
    @abc.abstractmethod
    def _resample_up_aperiodicity(feature, fs, new_fs, new_spectrum_len):
        raise NotImplementedError

    _resample_down_aperiodicity = _resample_down

    @classmethod
    def _resample(cls, feature, fs, new_fs, up_func, down_func):
        if fs == new_fs:
            return feature
        new_len = cls._resample_spectrum_len(feature, fs, new_fs)
        func = up_func if fs < new_fs else down_func
        return func(feature, fs, new_fs, new_len)

    @classmethod
    def resample_spectrum_envelope(cls, feature, fs, new_fs):
        return cls._resample(feature, fs, new_fs, cls._resample_up_spectrum_envelope,
                             cls._resample_down_spectrum_envelope)

    @classmethod
    def resample_aperiodicity(cls, feature, fs, new_fs):
        return cls._resample(feature, fs, new_fs, cls._resample_up_aperiodicity,
                             cls._resample_down_aperiodicity)

    @staticmethod
    @abc.abstractmethod
    def extract_is_voiced(feature):
        raise NotImplementedError

    # ---- silence ------------------------------------------------------------------------------
    @classmethod
    def create_silence_feature(cls, frame_len, fs, **kwargs):
        kwargs.setdefault('Synthesizer', cls)
        silence = kwiiyatta.feature(fs, **kwargs)
        silence.f0 = cls.silence_f0(frame_len, fs)
        silence.spectrum_envelope = cls.silence_spectrum_envelope(frame_len, fs)
        silence.aperiodicity = cls.silence_aperiodicity(frame_len, fs)
        return silence

    @staticmethod
    @abc.abstractmethod
    def silence_f0(frame_len, fs):
        raise NotImplementedError

    @staticmethod
    @abc.abstractmethod
    def _silence_spectrum_envelope(frame_len, fs, spectrum_len):
        raise NotImplementedError

    @classmethod
    def silence_spectrum_envelope(cls, frame_len, fs, spectrum_len=None):
        if spectrum_len is None:
            spectrum_len = cls.fs_spectrum_len(fs)
        return cls._silence_spectrum_envelope(frame_len, fs, spectrum_len)

    @staticmethod
    @abc.abstractmethod
    def _silence_aperiodicity(frame_len, fs, spectrum_len):
        raise NotImplementedError

    @classmethod
    def silence_aperiodicity(cls, frame_len, fs, spectrum_len=None):
        if spectrum_len is None:
            spectrum_len = cls.fs_spectrum_len(fs)
        return cls._silence_aperiodicity(frame_len, fs, spectrum_len)

    @staticmethod
    def silence_is_voiced(frame_len, fs):
        return np.full((frame_len), False)
