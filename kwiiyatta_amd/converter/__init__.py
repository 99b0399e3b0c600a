"""Converter layer (mirrors /root/reference/kwiiyatta/converter/__init__.py)."""
from . import abc
from .dataset import (AlignedDataset, ParallelDataset, TrimmedDataset, WavFileDataset,
                      make_dataset_to_array)
from .delta import DELTA_WINDOWS, DeltaFeatureConverter, DeltaFeatureDataset
from .gmm import GMMFeatureConverter
from .mcep import MelCepstrumDataset, MelCepstrumFeatureConverter


def MelCepstrumConverter(use_delta=True, mcep_fs=None, Converter=GMMFeatureConverter, **kwargs):
    converter = Converter(**kwargs)
    if use_delta:
        converter = DeltaFeatureConverter(converter)
    return MelCepstrumFeatureConverter(converter, mcep_fs=mcep_fs)


def align_dataset(parallel_dataset):
    return AlignedDataset(TrimmedDataset(parallel_dataset))


__all__ = ['MelCepstrumConverter', 'align_dataset', 'AlignedDataset', 'ParallelDataset',
           'TrimmedDataset', 'WavFileDataset', 'make_dataset_to_array', 'GMMFeatureConverter',
           'DELTA_WINDOWS', 'DeltaFeatureConverter', 'DeltaFeatureDataset', 'MelCepstrumDataset',
           'MelCepstrumFeatureConverter']
