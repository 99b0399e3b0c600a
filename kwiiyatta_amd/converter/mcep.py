"""Mel-cepstrum stage of the converter stack: feature sets become coefficient matrices c1..cN at one common order
and sampling rate for training; at conversion the power coefficient c0 is set aside and re-attached.  API of
kwiiyatta.converter.mcep (/root/reference/kwiiyatta/converter/mcep.py)."""
import copy

import numpy as np

from . import abc


def _pkg():
    import kwiiyatta_amd
    return kwiiyatta_amd


class MelCepstrumDataset(abc.MapDataset):
    """c1..cN of every item.  The first item fixes the order and (unless `mcep_fs` is given) the sampling rate;
    later items are truncated / resampled to them."""
    with_key = True

    def __init__(self, base, mcep_fs=None):
        super().__init__(base)
        self.fs, self.order = mcep_fs, None

    def function(self, feature, key):
        snap = _pkg().feature(feature)
        if self.order is None:
            self.order = snap.mel_cepstrum_order
        if self.fs is None:
            self.fs = snap.fs
        snap.mel_cepstrum_order = self.order
        record = snap.mel_cepstrum if snap.fs == self.fs else snap.resample_mel_cepstrum(self.fs)
        return record.data[:, 1:]


class MelCepstrumFeatureConverter(abc.MapFeatureConverter):
    def __init__(self, base, mcep_fs=None):
        super().__init__(base)
        self.mcep_fs = mcep_fs

    def train(self, dataset, keys, **kwargs):
        coefficients = MelCepstrumDataset(dataset, mcep_fs=self.mcep_fs)
        self.base.train(coefficients, keys, **kwargs)
        self.order, self.fs = coefficients.order, coefficients.fs

    def convert(self, mel_cepstrum, **kwargs):
        """a MelCepstrum at the converter's sampling rate: c0 of the input, c1..cN converted"""
        if mel_cepstrum.order != self.order:
            raise ValueError(f'order is expected to {self.order!s} but {mel_cepstrum.order!s}')
        out = copy.copy(mel_cepstrum) if mel_cepstrum.fs == self.fs else _pkg().resample(mel_cepstrum, self.fs)
        power, shape = out.data[:, :1], out.data[:, 1:]
        out.data = np.hstack((power, super().convert(shape, raw=mel_cepstrum, **kwargs)))
        return out
