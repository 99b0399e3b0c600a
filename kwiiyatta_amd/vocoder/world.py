"""WORLD vocoder backend of the feature model (mirrors
/root/reference/kwiiyatta/vocoder/world.py:13-166), running on the HIP kernels
through kwiiyatta_amd.backend.world (pyworld-shaped)."""
import numpy as np
import scipy.interpolate

import kwiiyatta_amd as kwiiyatta
from ..backend import world as pyworld
from ..wavfile import Wavdata
from . import abc


class WorldAnalyzer(abc.Analyzer):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._data = np.ascontiguousarray(self._data)
        self._timeaxis = None

    @property
    def frame_len(self):
        # integer arithmetic, as the reference: N*1000 // fs // frame_period + 1
        return self.data.shape[0] * 1000 // self.fs // self.frame_period + 1

    @property
    def spectrum_len(self):
        if self._spectrum_envelope is not None:
            return self._spectrum_envelope.shape[-1]
        return WorldSynthesizer.fs_spectrum_len(self.fs)

    def clear_features(self):
        super().clear_features()
        self._f0 = self._spectrum_envelope = self._aperiodicity = None

    def extract_f0(self, **kwargs):
        if self._f0 is None:
            f0, self._timeaxis = pyworld.dio(self.data, self.fs, frame_period=self.frame_period,
                                             **kwargs)
            self._f0 = pyworld.stonemask(self.data, f0, self._timeaxis, self.fs)
        return self._f0

    def extract_spectrum_envelope(self, **kwargs):
        if self._spectrum_envelope is None:
            # kwiiyatta stores the envelope divided by fs (world.py:49-50); the
            # division is folded into the kernel's epilogue.
            f0 = self.f0
            self._spectrum_envelope = pyworld.cheaptrick(self.data, f0, self._timeaxis, self.fs,
                                                         out_div=float(self.fs), **kwargs)
        return self._spectrum_envelope

    def extract_aperiodicity(self, **kwargs):
        if self._aperiodicity is None:
            f0 = self.f0
            self._aperiodicity = pyworld.d4c(self.data, f0, self._timeaxis, self.fs, **kwargs)
        return self._aperiodicity

    def ascontiguousarray(self):
        pass  # the kernels already produce C-contiguous float64 arrays


class WorldSynthesizer(abc.Synthesizer):
    SAFE_GUARD_MINIMUM = 0.000000000001
    EPS = 0.00000000000000022204460492503131
    FREQUENCY_INTERVAL = 3000.0
    UPPER_LIMIT = 15000.0

    @staticmethod
    def reshape_feature(feature):
        """WORLD synthesis needs 2^n + 1 bins: round the bin count up to that."""
        n = feature.spectrum_len
        pow2 = 1 << (n.bit_length() - 1)
        if pow2 < n - 1:
            pow2 *= 2
        return kwiiyatta.reshape(feature, pow2 + 1)

    @classmethod
    def _synthesize(cls, feature):
        f = cls.reshape_feature(feature)
        f.ascontiguousarray()
        # the stored envelope is 1/fs-scaled (world.py:88 multiplies it back);
        # the multiplication is folded into the kernel's spectrum load.
        y = pyworld.synthesize(f.f0, f.spectrum_envelope, f.aperiodicity, f.fs, f.frame_period,
                               sp_mul=float(f.fs))
        return Wavdata(f.fs, y)

    @staticmethod
    def fs_spectrum_len(fs):
        return pyworld.get_cheaptrick_fft_size(fs) // 2 + 1

    # ---- aperiodicity across sampling rates: through WORLD's band coding -----------------------
    @classmethod
    def _get_aperiodicity_num(cls, fs):
        return int(min(cls.UPPER_LIMIT, fs / 2 - cls.FREQUENCY_INTERVAL) / cls.FREQUENCY_INTERVAL)

    @classmethod
    def _reshape_aperiodicity(cls, feature, fs, new_spectrum_len):
        coded = pyworld.code_aperiodicity(np.ascontiguousarray(feature), fs)
        return pyworld.decode_aperiodicity(coded, fs, (new_spectrum_len - 1) * 2)

    @classmethod
    def _resample_up_spectrum_envelope(cls, feature, fs, new_fs, new_spectrum_len):
        extra = new_spectrum_len - feature.shape[1]
        return np.hstack((feature, cls.silence_spectrum_envelope(feature.shape[0], fs, extra)))

    @classmethod
    def _resample_down_aperiodicity(cls, feature, fs, new_fs, new_spectrum_len):
        coded = pyworld.code_aperiodicity(np.ascontiguousarray(feature), fs)
        num = cls._get_aperiodicity_num(new_fs)
        if num < coded.shape[1]:
            coded = np.ascontiguousarray(coded[:, :num])
        return pyworld.decode_aperiodicity(coded, new_fs, (new_spectrum_len - 1) * 2)

    @classmethod
    def _resample_up_aperiodicity(cls, feature, fs, new_fs, new_spectrum_len):
        coded = pyworld.code_aperiodicity(np.ascontiguousarray(feature), fs)
        num = cls._get_aperiodicity_num(new_fs)
        if num > coded.shape[1]:
            axis = np.hstack((np.arange(coded.shape[1]), new_fs / 2 / cls.FREQUENCY_INTERVAL - 1))
            coded = np.hstack((coded, np.full((coded.shape[0], 1), -cls.SAFE_GUARD_MINIMUM)))
            coded = np.ascontiguousarray(
                scipy.interpolate.interp1d(axis, coded, axis=1)(np.arange(num)))
        return pyworld.decode_aperiodicity(coded, new_fs, (new_spectrum_len - 1) * 2)

    @staticmethod
    def extract_is_voiced(feature):
        lowest_f0 = feature.fs / ((feature.spectrum_len - 1) / 2) + 1.0
        return np.logical_and(feature.f0 >= lowest_f0, feature.aperiodicity[:, 0] <= 0.999)

    @staticmethod
    def silence_f0(frame_len, fs):
        return np.zeros((frame_len))

    @classmethod
    def _silence_spectrum_envelope(cls, frame_len, fs, spectrum_len):
        # numpy's global legacy RNG on purpose: the reference's tests seed it
        return np.abs(np.random.normal(0, cls.EPS / fs, (frame_len, spectrum_len)))

    @classmethod
    def _silence_aperiodicity(cls, frame_len, fs, spectrum_len):
        return np.full((frame_len, spectrum_len), 1 - cls.SAFE_GUARD_MINIMUM)
