"""Waveform container and wav I/O (mirrors kwiiyatta.wavfile of the reference,
/root/reference/kwiiyatta/wavfile.py:8-53).  Host-side; no GPU work."""
import numpy as np
from scipy.io import wavfile as _scipy_wav


def normalize_data(data, peak_lv=-1):
    """Scale `data` in place so that its peak does not exceed 10**(peak_lv/10)."""
    limit = np.power(10, peak_lv / 10)
    peak = np.abs(data).max()
    if peak > limit:
        data *= limit / peak


class Wavdata:
    def __init__(self, fs, data):
        self.fs = fs
        self.data = data

    def normalize(self, peak_lv=-1):
        """Remove the DC offset, then (unless peak_lv is None) limit the peak."""
        self.data -= self.data.mean()
        if peak_lv is not None:
            normalize_data(self.data, peak_lv)

    def _as_int16(self):
        return (self.data * (2 ** 15)).astype(np.int16)

    def save(self, wav, normalize=True, **kwargs):
        if normalize:
            self.normalize(**kwargs)
        _scipy_wav.write(wav, self.fs, self._as_int16())

    def play(self, normalize=True, **kwargs):
        import pyaudio  # optional dependency, only needed for playback
        if normalize:
            self.normalize(**kwargs)
        audio = pyaudio.PyAudio()
        stream = audio.open(rate=self.fs, channels=1, format=pyaudio.paInt16, output=True)
        stream.write(self._as_int16(), num_frames=len(self.data))
        stream.close()
        audio.terminate()


def load_wav(wav):
    """u8 / i16 / i32 / f32 / f64 wav -> float64 in [-1, 1)."""
    fs, data = _scipy_wav.read(wav)
    kind = data.dtype.kind
    if kind == 'f':
        return Wavdata(fs, data.astype(np.float64))
    full_scale = 2 ** (data.dtype.itemsize * 8 - 1)
    if kind == 'u':
        assert data.dtype.itemsize == 1
        return Wavdata(fs, data.astype(np.float64) / full_scale - 1)
    assert kind == 'i'
    return Wavdata(fs, data.astype(np.float64) / full_scale)
