"""Mel-cepstrum container (mirrors kwiiyatta.vocoder.mcep.MelCepstrum,
/root/reference/kwiiyatta/vocoder/mcep.py:6-75).  spectrum <-> mel-cepstrum
runs on the GPU through kwiiyatta_amd.backend.sptk (pysptk-shaped)."""
import kwiiyatta_amd as kwiiyatta
from ..backend import sptk


class MelCepstrum:
    def __init__(self, fs, frame_period, data=None):
        self._fs = fs
        self._frame_period = frame_period
        self.data = data

    fs = property(lambda self: self._fs)
    frame_period = property(lambda self: self._frame_period)

    @property
    def order(self):
        return self.data.shape[-1] - 1

    @staticmethod
    def fs_alpha(fs):
        return sptk.mcepalpha(fs)

    def alpha(self):
        return self.fs_alpha(self.fs)

    def extract_spectrum(self, spectrum_len=None, Synthesizer=None):
        if spectrum_len is None:
            Synthesizer = Synthesizer or kwiiyatta.Synthesizer
            spectrum_len = Synthesizer.fs_spectrum_len(self.fs)
        return sptk.mc2sp(self.data, fftlen=(spectrum_len - 1) * 2, alpha=self.alpha())

    def extract_data(self, spectrum, order=24, fs=None):
        return sptk.sp2mc(spectrum, order=order, alpha=self.fs_alpha(self.fs if fs is None else fs))

    def extract(self, spectrum, order=24):
        self.data = self.extract_data(spectrum, order)
        return self.data

    def resample_data(self, new_fs, spectrum_len=None, Synthesizer=None, order=None):
        """Mel-cepstrum at another sampling rate, going through the spectrum."""
        Synthesizer = Synthesizer or kwiiyatta.Synthesizer
        if spectrum_len is None:
            spectrum_len = Synthesizer.fs_spectrum_len(self.fs)
        if order is None:
            order = self.order
        spec = Synthesizer.resample_spectrum_envelope(self.extract_spectrum(spectrum_len), self.fs, new_fs)
        return self.extract_data(spec, order, fs=new_fs)

    def resample(self, new_fs, spectrum=None, order=None, **kwargs):
        if order is None:
            order = self.order
        if spectrum is not None:
            self._fs = new_fs
            if order is not None:
                kwargs['order'] = order
            self.extract(spectrum, **kwargs)
        elif new_fs != self.fs:
            self.data = self.resample_data(new_fs, order=order, **kwargs)
            self._fs = new_fs
