"""Datasets of analysed wav files and their alignment (mirrors
/root/reference/kwiiyatta/converter/dataset.py:12-77)."""
import copy

import numpy as np

import kwiiyatta_amd as kwiiyatta
from . import abc


# nnmnkwii.preprocessing helpers the reference imports (dataset.py:3), restated:
def trim_zeros_frames(x, eps=1e-7):
    """Drop all-zero frames at the ends; NOTE the kept length is applied from
    the front (nnmnkwii behaviour the reference relies on, dataset.py:52)."""
    s = np.sum(np.abs(x), axis=1)
    s[s < eps] = 0.
    return x[:len(np.trim_zeros(s))]


def remove_zeros_frames(x, eps=1e-7):
    s = np.sum(np.abs(x), axis=1)
    s[s < eps] = 0.
    return x[s > eps]


class WavFileDataset(abc.Dataset):
    def __init__(self, data_dir, Analyzer=None):
        super().__init__()
        self.Analyzer = Analyzer if Analyzer is not None else kwiiyatta.analyze_wav
        self.data_dir = data_dir
        if not self.data_dir.exists():
            raise FileNotFoundError(f'wav files dir "{self.data_dir!s}" is not found')
        if not self.data_dir.is_dir():
            raise NotADirectoryError(f'wav files dir "{self.data_dir!s}" is not directory')
        self.files = frozenset(f.relative_to(self.data_dir) for f in self.data_dir.glob('*.wav'))

    def keys(self):
        return self.files

    def get_data(self, key):
        return self.Analyzer(self.data_dir / key)


class ParallelDataset(abc.Dataset):
    def __init__(self, dataset1, dataset2):
        super().__init__()
        self.dataset1 = dataset1
        self.dataset2 = dataset2
        self.common_keys = self.dataset1.keys() & self.dataset2.keys()

    def keys(self):
        return self.common_keys

    def get_data(self, key):
        return self.dataset1[key], self.dataset2[key]


@abc.map_dataset()
def TrimmedDataset(feature):
    kept = trim_zeros_frames(feature.spectrum_envelope)
    return feature[:len(kept)]


@abc.map_dataset(expand_tuple=False)
def AlignedDataset(features, **kwargs):
    a, b = features
    return kwiiyatta.align_even(a, b, **kwargs)


def make_dataset_to_array(dataset, keys=None):
    """Stack all items (tuples are joined column-wise) into one training matrix."""
    if keys is None:
        keys = sorted(dataset.keys())
    parts = []
    for key in keys:
        d = dataset[key]
        if isinstance(d, tuple):
            d = np.hstack(d)
        parts.append(remove_zeros_frames(d))
    if not parts:
        return None
    return np.concatenate(parts, axis=0) if len(parts) > 1 else copy.copy(parts[0])
