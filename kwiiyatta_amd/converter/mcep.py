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

    # ---- trained state on disk (an addition: the reference retrains on every run) ---------------------------------
    MODEL_FORMAT = 'kwiiyatta_amd.converter/1'

    def save(self, path):
        """the trained stack as one .npz: the mixture's parameters and what the outer stages learnt from the
        training set (mel-cepstrum order, sampling rate, frame period)"""
        gmm = self.gmm
        with open(path, 'wb') as fh:        # a file object: np.savez would append '.npz' to a bare name
            np.savez(fh, format=self.MODEL_FORMAT, order=self.order, fs=self.fs,
                     frame_period=getattr(self, 'frame_period', -1),     # (forwarded to the delta stage)
                     weights=gmm.weights_, means=gmm.means_, covariances=gmm.covariances_)

    def load(self, path):
        """the state written by `save` into this (untrained) stack; component count and dimensions come from
        the file"""
        with np.load(path, allow_pickle=False) as z:
            if str(z['format']) != self.MODEL_FORMAT:
                raise ValueError(f'{path!s}: not a converter model of format {self.MODEL_FORMAT}')
            self.order, self.fs = int(z['order']), int(z['fs'])
            from .delta import DeltaFeatureConverter
            stage = self.base
            while isinstance(stage, abc.MapFeatureConverter):
                if isinstance(stage, DeltaFeatureConverter):       # it keeps the training set's frame period
                    period = float(z['frame_period'])
                    stage.frame_period = int(period) if period.is_integer() else period
                stage = stage.base
            gmm = self.gmm
            gmm.weights_ = np.array(z['weights'], dtype=np.float64)
            gmm.means_ = np.array(z['means'], dtype=np.float64)
            gmm.covariances_ = np.array(z['covariances'], dtype=np.float64)
            gmm.n_components = len(gmm.weights_)
            gmm.converged_ = True
        return self

    def convert(self, mel_cepstrum, **kwargs):
        """a MelCepstrum at the converter's sampling rate: c0 of the input, c1..cN converted"""
        if mel_cepstrum.order != self.order:
            raise ValueError(f'order is expected to {self.order!s} but {mel_cepstrum.order!s}')
        out = copy.copy(mel_cepstrum) if mel_cepstrum.fs == self.fs else _pkg().resample(mel_cepstrum, self.fs)
        power, shape = out.data[:, :1], out.data[:, 1:]
        out.data = np.hstack((power, super().convert(shape, raw=mel_cepstrum, **kwargs)))
        return out
