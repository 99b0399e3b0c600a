"""Mel-cepstrum stage of the converter stack: drops/re-attaches the power
coefficient and unifies sampling rate and order (mirrors
/root/reference/kwiiyatta/converter/mcep.py:10-61)."""
import copy

import numpy as np

import kwiiyatta_amd as kwiiyatta
from . import abc


class MelCepstrumDataset(abc.MapDataset):
    with_key = True

    def __init__(self, base, mcep_fs=None):
        super().__init__(base)
        self.fs = mcep_fs
        self.order = None

    def function(self, feature, key):
        f = kwiiyatta.feature(feature)
        if self.order is None:
            self.order = f.mel_cepstrum_order
        elif self.order != feature.mel_cepstrum_order:
            f.mel_cepstrum_order = self.order
        mcep = f.mel_cepstrum.data
        if self.fs is None:
            self.fs = f.fs
        elif self.fs != f.fs:
            mcep = f.resample_mel_cepstrum(self.fs).data
        return mcep[:, 1:]          # the power coefficient is not converted


class MelCepstrumFeatureConverter(abc.MapFeatureConverter):
    def __init__(self, base, mcep_fs=None):
        super().__init__(base)
        self.mcep_fs = mcep_fs

    def train(self, dataset, keys, **kwargs):
        mcep_dataset = MelCepstrumDataset(dataset, mcep_fs=self.mcep_fs)
        self.base.train(mcep_dataset, keys, **kwargs)
        self.order = mcep_dataset.order
        self.fs = mcep_dataset.fs

    def convert(self, mel_cepstrum, **kwargs):
        if self.order != mel_cepstrum.order:
            raise ValueError(f'order is expected to {self.order!s} but {mel_cepstrum.order!s}')
        if self.fs != mel_cepstrum.fs:
            result = kwiiyatta.resample(mel_cepstrum, self.fs)
        else:
            result = copy.copy(mel_cepstrum)
        converted = super().convert(result.data[:, 1:], raw=mel_cepstrum, **kwargs)
        result.data = np.hstack((result.data[:, 0].reshape(-1, 1), converted))
        return result
