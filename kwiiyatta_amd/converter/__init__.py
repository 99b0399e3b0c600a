"""Converter layer: datasets of the training path and the converter stack
(mel-cepstrum stage > delta stage > joint GMM).  Export list of kwiiyatta.converter."""
from . import abc
from .dataset import AlignedDataset, ParallelDataset, TrimmedDataset, WavFileDataset, make_dataset_to_array
from .delta import DELTA_WINDOWS, DeltaFeatureConverter, DeltaFeatureDataset
from .gmm import GMMFeatureConverter
from .mcep import MelCepstrumDataset, MelCepstrumFeatureConverter


def MelCepstrumConverter(use_delta=True, mcep_fs=None, Converter=GMMFeatureConverter, **kwargs):
    """the stack the CLIs use; kwargs go to the innermost converter"""
    stack = Converter(**kwargs)
    if use_delta:
        stack = DeltaFeatureConverter(stack)
    return MelCepstrumFeatureConverter(stack, mcep_fs=mcep_fs)


def align_dataset(parallel_dataset):
    """a parallel dataset -> trimmed and aligned pairs (the training set of the converter)"""
    return AlignedDataset(TrimmedDataset(parallel_dataset))


__all__ = ['MelCepstrumConverter', 'align_dataset', 'AlignedDataset', 'ParallelDataset', 'TrimmedDataset',
           'WavFileDataset', 'make_dataset_to_array', 'GMMFeatureConverter', 'DELTA_WINDOWS',
           'DeltaFeatureConverter', 'DeltaFeatureDataset', 'MelCepstrumDataset', 'MelCepstrumFeatureConverter']
