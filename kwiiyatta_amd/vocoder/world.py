"""WORLD as the vocoder of the feature model: `WorldAnalyzer` fills the slots with DIO + StoneMask, CheapTrick and
D4C, `WorldSynthesizer` turns feature sets back into waveforms and carries WORLD's conventions (bin count per
sampling rate, band-coded aperiodicity, voicing rule, silence).  API of kwiiyatta.vocoder.world
(/root/reference/kwiiyatta/vocoder/world.py:13-166); the signal processing runs in the HIP kernels behind
kwiiyatta_amd.backend.world, which has pyworld's call signatures.

Two conventions of the reference are folded into kernel arguments instead of separate array passes: the envelope
is stored divided by the sampling rate (world.py:50; `out_div`) and multiplied back for synthesis (world.py:88;
`sp_mul`)."""
import numpy as np
import scipy.interpolate

from ..backend import world as pyworld
from ..wavfile import Wavdata
from . import abc


class WorldAnalyzer(abc.Analyzer):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._data = np.ascontiguousarray(self._data)      # pyworld's contract: C-contiguous float64
        self._timeaxis = None

    @property
    def frame_len(self):
        samples = self.data.shape[0]
        return samples * 1000 // self.fs // self.frame_period + 1      # integer arithmetic throughout

    def extract_f0(self, **kwargs):
        if self._f0 is None:
            coarse, self._timeaxis = pyworld.dio(self.data, self.fs, frame_period=self.frame_period, **kwargs)
            self._f0 = pyworld.stonemask(self.data, coarse, self._timeaxis, self.fs)
        return self._f0

    def _frame_grid(self):
        """(f0, frame times): the spectral analyses run on the f0 analysis' frames"""
        return self.f0, self._timeaxis

    def extract_spectrum_envelope(self, **kwargs):
        if self._spectrum_envelope is None:
            f0, times = self._frame_grid()
            self._spectrum_envelope = pyworld.cheaptrick(self.data, f0, times, self.fs, out_div=float(self.fs),
                                                         **kwargs)
        return self._spectrum_envelope

    def extract_aperiodicity(self, **kwargs):
        if self._aperiodicity is None:
            f0, times = self._frame_grid()
            self._aperiodicity = pyworld.d4c(self.data, f0, times, self.fs, **kwargs)
        return self._aperiodicity

    def ascontiguousarray(self):
        pass        # what the kernels return is contiguous already


class WorldSynthesizer(abc.Synthesizer):
    SAFE_GUARD_MINIMUM = 0.000000000001
    EPS = 0.00000000000000022204460492503131
    FREQUENCY_INTERVAL = 3000.0
    UPPER_LIMIT = 15000.0

    @staticmethod
    def fs_spectrum_len(fs):
        return pyworld.get_cheaptrick_fft_size(fs) // 2 + 1

    # ---- waveform -------------------------------------------------------------------------------------------
    @staticmethod
    def reshape_feature(feature):
        """WORLD synthesises from 2^n + 1 bins: a feature set with any other count (after a change of sampling
        rate) is stretched to the next such count"""
        import kwiiyatta_amd
        span = feature.spectrum_len - 1
        pow2 = max(2, 1 << (span - 1).bit_length())          # the smallest power of two >= span
        return kwiiyatta_amd.reshape(feature, pow2 + 1)

    @classmethod
    def _synthesize(cls, feature):
        f = cls.reshape_feature(feature)
        f.ascontiguousarray()
        samples = pyworld.synthesize(f.f0, f.spectrum_envelope, f.aperiodicity, f.fs, f.frame_period,
                                     sp_mul=float(f.fs))
        return Wavdata(f.fs, samples)

    # ---- aperiodicity between sampling rates: through WORLD's band code (one dB value per 3 kHz band) ---------
    @classmethod
    def _get_aperiodicity_num(cls, fs):
        return int(min(cls.UPPER_LIMIT, fs / 2 - cls.FREQUENCY_INTERVAL) / cls.FREQUENCY_INTERVAL)

    @classmethod
    def _recode_aperiodicity(cls, feature, fs, new_fs, new_spectrum_len):
        bands = pyworld.code_aperiodicity(np.ascontiguousarray(feature), fs)
        have, want = bands.shape[1], cls._get_aperiodicity_num(new_fs)
        if want < have:
            bands = np.ascontiguousarray(bands[:, :want])
        elif want > have:
            # extend towards the new Nyquist frequency, where WORLD pins the contour at -1e-12 dB
            knots = np.append(np.arange(have), new_fs / 2 / cls.FREQUENCY_INTERVAL - 1)
            values = np.hstack((bands, np.full((len(bands), 1), -cls.SAFE_GUARD_MINIMUM)))
            bands = np.ascontiguousarray(scipy.interpolate.interp1d(knots, values, axis=1)(np.arange(want)))
        return pyworld.decode_aperiodicity(bands, new_fs, 2 * (new_spectrum_len - 1))

    _resample_up_aperiodicity = _recode_aperiodicity
    _resample_down_aperiodicity = _recode_aperiodicity

    @classmethod
    def _resample_up_spectrum_envelope(cls, feature, fs, new_fs, new_spectrum_len):
        """the band above the old Nyquist frequency is silence"""
        missing = new_spectrum_len - feature.shape[1]
        return np.hstack((feature, cls.silence_spectrum_envelope(len(feature), fs, missing)))

    # ---- voicing, silence ----------------------------------------------------------------------------------------
    @staticmethod
    def extract_is_voiced(feature):
        floor = feature.fs / ((feature.spectrum_len - 1) / 2) + 1.0      # kwiiyatta's own threshold, not WORLD's
        return np.logical_and(feature.f0 >= floor, feature.aperiodicity[:, 0] <= 0.999)

    @staticmethod
    def silence_f0(frame_len, fs):
        return np.zeros(frame_len)

    @classmethod
    def _silence_spectrum_envelope(cls, frame_len, fs, spectrum_len):
        # the reference draws from numpy's GLOBAL generator (its tests seed it): kept
        return np.abs(np.random.normal(0, cls.EPS / fs, (frame_len, spectrum_len)))

    @classmethod
    def _silence_aperiodicity(cls, frame_len, fs, spectrum_len):
        return np.full((frame_len, spectrum_len), 1 - cls.SAFE_GUARD_MINIMUM)
