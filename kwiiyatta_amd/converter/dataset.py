"""Datasets of the training path: wav directories, parallel pairs, trimming, alignment, and the stacking of a
dataset into the matrix the converter is fitted on.  API of kwiiyatta.converter.dataset
(/root/reference/kwiiyatta/converter/dataset.py).  `kwiiyatta_amd.corpus` is the device-resident counterpart of
this chain for whole corpora; this module is the per-item Python path (and the definition the other is tested
against)."""
import numpy as np

from . import abc

FRAME_EPS = 1e-7        # nnmnkwii: a frame whose L1 norm is below this is "zero"


def _pkg():
    import kwiiyatta_amd
    return kwiiyatta_amd


def _frame_mass(x, eps):
    mass = np.abs(x).sum(axis=1)
    mass[mass < eps] = 0.
    return mass


def trim_zeros_frames(x, eps=FRAME_EPS):
    """nnmnkwii.preprocessing.trim_zeros_frames: the zero frames at BOTH ends are counted off, but the remaining
    length is taken from the FRONT (x[:n]) -- behaviour the reference inherits and documents with a strict xfail
    (tests/kwiiyatta/test_dataset.py:78-99)"""
    return x[:len(np.trim_zeros(_frame_mass(x, eps)))]


def remove_zeros_frames(x, eps=FRAME_EPS):
    """nnmnkwii.preprocessing.remove_zeros_frames: all zero frames are dropped"""
    return x[_frame_mass(x, eps) > 0]


class WavFileDataset(abc.Dataset):
    """the *.wav files of a directory, analysed (lazily) on access; keys are paths relative to the directory"""

    def __init__(self, data_dir, Analyzer=None):
        if not data_dir.exists():
            raise FileNotFoundError(f'wav files dir "{data_dir!s}" is not found')
        if not data_dir.is_dir():
            raise NotADirectoryError(f'wav files dir "{data_dir!s}" is not directory')
        self.data_dir = data_dir
        self.Analyzer = Analyzer or _pkg().analyze_wav
        self.files = frozenset(path.relative_to(data_dir) for path in data_dir.glob('*.wav'))

    def keys(self):
        return self.files

    def get_data(self, key):
        return self.Analyzer(self.data_dir / key)


class ParallelDataset(abc.Dataset):
    """(item of dataset1, item of dataset2) for the keys both have"""

    def __init__(self, dataset1, dataset2):
        self.dataset1, self.dataset2 = dataset1, dataset2
        self.common_keys = dataset1.keys() & dataset2.keys()

    def keys(self):
        return self.common_keys

    def get_data(self, key):
        return self.dataset1[key], self.dataset2[key]


@abc.map_dataset()
def TrimmedDataset(feature):
    """a feature set without its silent end frames (see trim_zeros_frames for which ones exactly)"""
    return feature[:len(trim_zeros_frames(feature.spectrum_envelope))]


@abc.map_dataset(expand_tuple=False)
def AlignedDataset(features, **kwargs):
    """a parallel pair, both sides along their common warping path"""
    return _pkg().align_even(*features, **kwargs)


def make_dataset_to_array(dataset, keys=None):
    """rows of all items under `keys` (default: all, sorted), zero rows removed; a tuple item contributes its
    members side by side.  None for no keys."""
    blocks = []
    for key in (sorted(dataset.keys()) if keys is None else keys):
        item = dataset[key]
        blocks.append(remove_zeros_frames(np.hstack(item) if isinstance(item, tuple) else item))
    return np.concatenate(blocks) if blocks else None
