"""The feature model of the vocoder layer: what a per-utterance feature set IS, independent of the vocoder that
computes it.  Public surface of the reference's kwiiyatta.vocoder.abc (Feature, MutableFeature, Analyzer,
Synthesizer; /root/reference/kwiiyatta/vocoder/abc/), rebuilt around one mechanism:

    A feature set has four array slots -- f0 (T,), spectrum_envelope (T, K), aperiodicity (T, K),
    is_voiced (T,) -- and a MelCepstrum record.  A slot is either filled, or empty with a *provider*: the name
    of a method that fills it on first use.  Materialised features (`vocoder.feature.Feature`) have no
    providers and get their slots assigned; analyzers declare providers that run the vocoder on their waveform.
    Everything else -- the spectrum / mel-cepstrum duality, slicing, comparison, change of sampling rate or
    bin count -- is written once against `_peek` (the slot as it is) and `_slot` (the slot, filled if possible).

The slots are plain attributes (`_f0`, `_spectrum_envelope`, `_aperiodicity`, `_is_voiced`, `_mel_cepstrum`)
because the reference's tests reach into them (tests/kwiiyatta/test_vocoder.py:22-110, 291-467).
All arithmetic is delegated: spectra <-> mel-cepstra to `MelCepstrum` (GPU), the spectral-axis operations to
`vocoder.spectral` through the Synthesizer class of the feature.
"""
import abc
import copy

import numpy as np

from .. import spectral
from ..mcep import MelCepstrum

__all__ = ['Analyzer', 'Feature', 'MutableFeature', 'Synthesizer']

ARRAY_SLOTS = ('f0', 'spectrum_envelope', 'aperiodicity')


def _pkg():
    import kwiiyatta_amd
    return kwiiyatta_amd


class Feature(abc.ABC):
    _providers = {}                                   # slot -> name of the method that fills it
    _width_slots = ('spectrum_envelope', 'aperiodicity')   # slots whose width decides `spectrum_len`

    def __init__(self, fs, frame_period=5, mcep_order=24, Synthesizer=None):
        self.mel_cepstrum_order = mcep_order
        self.Synthesizer = _pkg().Synthesizer if Synthesizer is None else Synthesizer
        self._mel_cepstrum = MelCepstrum(fs, frame_period)
        self._f0 = self._spectrum_envelope = self._aperiodicity = self._is_voiced = None

    # ---- slots -----------------------------------------------------------------------------------------
    def _peek(self, slot):
        return getattr(self, '_' + slot)

    def _put(self, slot, value):
        setattr(self, '_' + slot, value)

    def _slot(self, slot):
        value = self._peek(slot)
        if value is None and slot in self._providers:
            value = getattr(self, self._providers[slot])()
        return value

    # the reference's internal accessors, kept for code written against them
    def _get_f0(self):
        return self._slot('f0')

    def _get_spectrum_envelope(self):
        return self._slot('spectrum_envelope')

    def _get_aperiodicity(self):
        return self._slot('aperiodicity')

    def __copy__(self):
        twin = object.__new__(type(self))
        twin.__dict__.update(self.__dict__)
        twin._mel_cepstrum = copy.copy(self._mel_cepstrum)      # the record is per feature, its array is shared
        return twin

    # ---- scalars ------------------------------------------------------------------------------------------
    fs = property(lambda self: self._mel_cepstrum.fs)
    frame_period = property(lambda self: self._mel_cepstrum.frame_period)

    @property
    def frame_len(self):
        return self.f0.shape[0]

    @property
    def spectrum_len(self):
        for slot in self._width_slots:
            held = self._peek(slot)
            if held is not None:
                return held.shape[-1]
        return self.Synthesizer.fs_spectrum_len(self.fs)

    # ---- the arrays ------------------------------------------------------------------------------------------
    @property
    def f0(self):
        return self._slot('f0')

    @property
    def aperiodicity(self):
        return self._slot('aperiodicity')

    @property
    def spectrum_envelope(self):
        held = self._slot('spectrum_envelope')
        return held if held is not None else self._spectrum_from_mel_cepstrum()

    def _spectrum_from_mel_cepstrum(self, spectrum_len=None):
        if self._mel_cepstrum.data is None:
            return None
        return self._mel_cepstrum.extract_spectrum(self.spectrum_len if spectrum_len is None else spectrum_len)

    def extract_spectrum_envelope(self, spectrum_len=None):
        """the envelope implied by the stored mel-cepstrum (None without one)"""
        return self._spectrum_from_mel_cepstrum(spectrum_len)

    @property
    def is_voiced(self):
        return self.extract_is_voiced()

    def extract_is_voiced(self):
        if self._is_voiced is None:
            self._is_voiced = self.Synthesizer.extract_is_voiced(self)
        return self._is_voiced

    # ---- mel-cepstrum ---------------------------------------------------------------------------------------------
    @property
    def mel_cepstrum(self):
        return self.extract_mel_cepstrum()

    def extract_mel_cepstrum(self, spectrum=None):
        """The MelCepstrum of order `mel_cepstrum_order`.  A stored one of that order is returned as it is, a
        stored one of higher order is truncated (into a copy); otherwise it is computed -- from `spectrum` if
        given (which then replaces the stored envelope as the source of truth), else from the envelope."""
        record, want = self._mel_cepstrum, self.mel_cepstrum_order
        if spectrum is not None:
            self._put('spectrum_envelope', None)
        elif record.data is not None and record.order >= want:
            if record.order == want:
                return record
            low = copy.copy(record)
            low.data = record.data[:, :want + 1]
            return low
        source = spectrum if spectrum is not None else self.spectrum_envelope
        if source is None:
            return None
        record.extract(source, want)
        return record

    def clear_mel_cepstrum(self):
        self._mel_cepstrum.data = None

    # ---- other bin counts / sampling rates (results, not in place) ----------------------------------------------
    def reshaped_spectrum_envelope(self, new_spectrum_len):
        held = self._slot('spectrum_envelope')
        if held is None:                      # mel-cepstrum only: evaluate it on the new grid directly
            return self._mel_cepstrum.extract_spectrum(new_spectrum_len)
        return self.Synthesizer.reshape_spectrum_envelope(held, self.fs, new_spectrum_len)

    def reshaped_aperiodicity(self, new_spectrum_len):
        return self.Synthesizer.reshape_aperiodicity(self.aperiodicity, self.fs, new_spectrum_len)

    def resample_spectrum_envelope(self, new_fs):
        return self.Synthesizer.resample_spectrum_envelope(self.spectrum_envelope, self.fs, new_fs)

    def resample_aperiodicity(self, new_fs):
        return self.Synthesizer.resample_aperiodicity(self.aperiodicity, self.fs, new_fs)

    def resample_mel_cepstrum(self, new_fs):
        how = dict(order=self.mel_cepstrum_order)
        if self._slot('spectrum_envelope') is not None:
            how['spectrum'] = self.resample_spectrum_envelope(new_fs)
        else:
            how['Synthesizer'] = self.Synthesizer
        return _pkg().resample(self.mel_cepstrum, new_fs, **how)

    def synthesize(self, **kwargs):
        return self.Synthesizer.synthesize(self, **kwargs)

    @abc.abstractmethod
    def ascontiguousarray(self):
        raise NotImplementedError

    # ---- comparison, indexing ----------------------------------------------------------------------------------------
    def __eq__(self, other):
        if (self.fs, self.frame_period) != (other.fs, other.frame_period):
            return False
        return all(not (self._slot(s) != other._slot(s)).any() for s in ARRAY_SLOTS)

    __hash__ = None

    def __getitem__(self, key):
        """one frame (int) -> (f0, spectrum, aperiodicity, mel-cepstrum) of it; anything else numpy accepts along
        the frame axis -> a materialised feature of those frames (a stored mel-cepstrum / voicing follows)"""
        rows = {slot: getattr(self, slot)[key] for slot in ARRAY_SLOTS}
        stored_mcep = self._mel_cepstrum.data is not None
        mcep = self.mel_cepstrum.data[key] if stored_mcep or isinstance(key, int) else None
        if isinstance(key, int):
            return rows['f0'], rows['spectrum_envelope'], rows['aperiodicity'], mcep
        part = _pkg().feature(self)
        for slot, value in rows.items():
            part._put(slot, value)
        if self._is_voiced is not None:
            part._is_voiced = self._is_voiced[key]
        if mcep is not None:
            part._mel_cepstrum.data = mcep
        return part


class MutableFeature(Feature):
    """Slots can be assigned; assignments keep the derived data honest: new f0 / envelope / aperiodicity void
    the voicing decision, a new envelope voids the mel-cepstrum, dropping the envelope (None) first saves it as
    a mel-cepstrum, and a new mel-cepstrum takes over from the envelope."""

    # the reference's internal mutators
    def _set_f0(self, value):
        self._put('f0', value)

    def _set_spectrum_envelope(self, value):
        self._put('spectrum_envelope', value)

    def _set_aperiodicity(self, value):
        self._put('aperiodicity', value)

    def _assign(self, slot, value):
        if value is not None:
            self._is_voiced = None
        self._put(slot, value)

    @Feature.f0.setter
    def f0(self, value):
        self._assign('f0', value)

    @Feature.aperiodicity.setter
    def aperiodicity(self, value):
        self._assign('aperiodicity', value)

    @Feature.spectrum_envelope.setter
    def spectrum_envelope(self, value):
        if value is not None:
            self._mel_cepstrum.data = None
        elif self._mel_cepstrum.data is None:
            self.extract_mel_cepstrum()
        self._assign('spectrum_envelope', value)

    @Feature.mel_cepstrum.setter
    def mel_cepstrum(self, value):
        if isinstance(value, MelCepstrum):
            value = value.data if value.fs == self.fs else value.resample_data(self.fs, Synthesizer=self.Synthesizer)
        elif value is not None and not isinstance(value, np.ndarray):
            raise TypeError('Feature.mel_cepstrum should be a MelCepstrum or ndarray')
        if value is not None:
            self._put('spectrum_envelope', None)
        self._mel_cepstrum.data = value

    def reshape(self, new_spectrum_len):
        for slot, fresh in (('spectrum_envelope', self.reshaped_spectrum_envelope),
                            ('aperiodicity', self.reshaped_aperiodicity)):
            if self._slot(slot).shape[1] != new_spectrum_len:
                self._put(slot, fresh(new_spectrum_len))

    def resample(self, new_fs):
        if new_fs == self.fs:
            return
        # f0 does not depend on the sampling rate
        if self._slot('aperiodicity') is not None:
            self._put('aperiodicity', self.resample_aperiodicity(new_fs))
        if self._slot('spectrum_envelope') is not None:
            self._put('spectrum_envelope', self.resample_spectrum_envelope(new_fs))
            self._mel_cepstrum.data = None
        elif self._mel_cepstrum.data is not None:
            self._mel_cepstrum.resample(new_fs, Synthesizer=self.Synthesizer)
        self._mel_cepstrum._fs = new_fs

    def ascontiguousarray(self):
        for slot in ARRAY_SLOTS:
            self._put(slot, np.ascontiguousarray(getattr(self, slot)))


class Analyzer(Feature):
    """A feature set computed on demand from a waveform: every array slot has a provider (`extract_<slot>`),
    which a concrete vocoder implements and which caches its result in the slot."""
    _providers = {slot: 'extract_' + slot for slot in ARRAY_SLOTS}
    _width_slots = ('spectrum_envelope',)

    def __init__(self, wavdata, **kwargs):
        super().__init__(wavdata.fs, **kwargs)
        self._data = wavdata.data

    @classmethod
    def load_wav(cls, wavfile, **kwargs):
        return cls(_pkg().load_wav(wavfile), **kwargs)

    data = property(lambda self: self._data)

    @property
    def wavdata(self):
        return _pkg().Wavdata(self.fs, self.data)

    @property
    @abc.abstractmethod
    def frame_len(self):
        raise NotImplementedError

    @abc.abstractmethod
    def extract_f0(self):
        raise NotImplementedError

    @abc.abstractmethod
    def extract_spectrum_envelope(self):
        raise NotImplementedError

    @abc.abstractmethod
    def extract_aperiodicity(self):
        raise NotImplementedError

    def clear_features(self):
        self.clear_mel_cepstrum()
        for slot in ARRAY_SLOTS:
            self._put(slot, None)

    def ascontiguousarray(self):
        for slot in ARRAY_SLOTS:
            self._put(slot, np.ascontiguousarray(getattr(self, slot)))


class Synthesizer(abc.ABC):
    """What a vocoder contributes besides analysis, as class-level functions: the waveform of a feature set, its
    native number of bins per sampling rate, how its features move between bin counts and sampling rates, its
    voicing rule and its idea of silence."""

    # ---- waveform ------------------------------------------------------------------------------------------------
    @classmethod
    def synthesize(cls, feature, normalize=True, **kwargs):
        wavdata = cls._synthesize(feature)
        if normalize:
            cls.finish(wavdata, feature.frame_len, **kwargs)
        return wavdata

    @staticmethod
    def finish(wavdata, frame_len, **kwargs):
        """the post-step of `synthesize` on a raw waveform (also applied to the batch path's waveforms)"""
        wavdata.normalize(None)                          # DC only
        # peak limiting in 1 ms pieces -- as many pieces as the feature has FRAMES (the reference's loop)
        fs, limit = wavdata.fs, _pkg().wavfile.normalize_data
        for piece in range(frame_len):
            limit(wavdata.data[fs * piece // 1000:fs * (piece + 1) // 1000], **kwargs)
        return wavdata

    @staticmethod
    @abc.abstractmethod
    def _synthesize(feature):
        raise NotImplementedError

    @staticmethod
    def fs_spectrum_len(fs):
        raise NotImplementedError

    @staticmethod
    @abc.abstractmethod
    def extract_is_voiced(feature):
        raise NotImplementedError

    # ---- bin count ---------------------------------------------------------------------------------------------------
    @classmethod
    def reshape_spectrum_envelope(cls, feature, fs, new_spectrum_len):
        return spectral.stretch_log(feature, new_spectrum_len)

    @classmethod
    def reshape_aperiodicity(cls, feature, fs, new_spectrum_len):
        return spectral.stretch_log(feature, new_spectrum_len)

    # ---- sampling rate: vocoders say how a feature grows; by default it shrinks by losing its upper bins ------------
    @staticmethod
    @abc.abstractmethod
    def _resample_up_spectrum_envelope(feature, fs, new_fs, new_spectrum_len):
        raise NotImplementedError

    @staticmethod
    @abc.abstractmethod
    def _resample_up_aperiodicity(feature, fs, new_fs, new_spectrum_len):
        raise NotImplementedError

    @staticmethod
    def _resample_down_spectrum_envelope(feature, fs, new_fs, new_spectrum_len):
        return spectral.keep_low_band(feature, new_spectrum_len)

    @staticmethod
    def _resample_down_aperiodicity(feature, fs, new_fs, new_spectrum_len):
        return spectral.keep_low_band(feature, new_spectrum_len)

    @classmethod
    def resample_spectrum_envelope(cls, feature, fs, new_fs):
        return spectral.change_rate(
            feature, fs, new_fs, cls._resample_up_spectrum_envelope,
            lambda rows, bins: cls._resample_down_spectrum_envelope(rows, fs, new_fs, bins))

    @classmethod
    def resample_aperiodicity(cls, feature, fs, new_fs):
        return spectral.change_rate(
            feature, fs, new_fs, cls._resample_up_aperiodicity,
            lambda rows, bins: cls._resample_down_aperiodicity(rows, fs, new_fs, bins))

    # ---- silence -------------------------------------------------------------------------------------------------------
    @staticmethod
    @abc.abstractmethod
    def silence_f0(frame_len, fs):
        raise NotImplementedError

    @staticmethod
    @abc.abstractmethod
    def _silence_spectrum_envelope(frame_len, fs, spectrum_len):
        raise NotImplementedError

    @staticmethod
    @abc.abstractmethod
    def _silence_aperiodicity(frame_len, fs, spectrum_len):
        raise NotImplementedError

    @classmethod
    def _bins(cls, fs, spectrum_len):
        return cls.fs_spectrum_len(fs) if spectrum_len is None else spectrum_len

    @classmethod
    def silence_spectrum_envelope(cls, frame_len, fs, spectrum_len=None):
        return cls._silence_spectrum_envelope(frame_len, fs, cls._bins(fs, spectrum_len))

    @classmethod
    def silence_aperiodicity(cls, frame_len, fs, spectrum_len=None):
        return cls._silence_aperiodicity(frame_len, fs, cls._bins(fs, spectrum_len))

    @staticmethod
    def silence_is_voiced(frame_len, fs):
        return np.zeros(frame_len, dtype=bool)

    @classmethod
    def create_silence_feature(cls, frame_len, fs, **kwargs):
        quiet = _pkg().feature(fs, **{'Synthesizer': cls, **kwargs})
        quiet.f0 = cls.silence_f0(frame_len, fs)
        quiet.spectrum_envelope = cls.silence_spectrum_envelope(frame_len, fs)
        quiet.aperiodicity = cls.silence_aperiodicity(frame_len, fs)
        return quiet
