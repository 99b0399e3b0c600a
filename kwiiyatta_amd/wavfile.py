"""Waveforms in memory (`Wavdata`: sampling rate + float64 samples in [-1, 1)) and on disk.  API of kwiiyatta.wavfile
(/root/reference/kwiiyatta/wavfile.py).  Host-side only."""
import numpy as np
from scipy.io import wavfile as _wav

INT16_SCALE = 2 ** 15


def normalize_data(data, peak_lv=-1):
    """in place: scale down so that the peak is at most 10^(peak_lv / 10) (the reference's power-style dB)"""
    ceiling, peak = np.power(10, peak_lv / 10), np.abs(data).max()
    if peak > ceiling:
        data *= ceiling / peak


class Wavdata:
    def __init__(self, fs, data):
        self.fs, self.data = fs, data

    def normalize(self, peak_lv=-1):
        """in place: remove the mean; limit the peak unless peak_lv is None"""
        self.data -= self.data.mean()
        if peak_lv is not None:
            normalize_data(self.data, peak_lv)

    def _pcm16(self, normalize, options):
        if normalize:
            self.normalize(**options)
        return (self.data * INT16_SCALE).astype(np.int16)        # truncation, as the reference

    def save(self, wav, normalize=True, **kwargs):
        _wav.write(wav, self.fs, self._pcm16(normalize, kwargs))

    def play(self, normalize=True, **kwargs):
        import pyaudio                                          # optional: only playback needs it
        pcm = self._pcm16(normalize, kwargs)
        audio = pyaudio.PyAudio()
        try:
            out = audio.open(rate=self.fs, channels=1, format=pyaudio.paInt16, output=True)
            out.write(pcm, num_frames=len(pcm))
            out.close()
        finally:
            audio.terminate()


def load_wav(wav):
    """8-bit unsigned, 16/32-bit signed and 32/64-bit float wav files -> Wavdata with float64 samples"""
    fs, samples = _wav.read(wav)
    kind, bits = samples.dtype.kind, samples.dtype.itemsize * 8
    as_f64 = samples.astype(np.float64)
    if kind == 'f':
        return Wavdata(fs, as_f64)
    half_range = 2 ** (bits - 1)
    if kind == 'u':
        assert bits == 8
        return Wavdata(fs, as_f64 / half_range - 1)
    assert kind == 'i'
    return Wavdata(fs, as_f64 / half_range)
