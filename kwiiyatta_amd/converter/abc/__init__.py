"""Abstract bases of the converter layer (mirrors
/root/reference/kwiiyatta/converter/abc/{converter,dataset}.py): lazily
evaluated Mapping-style datasets that can be stacked as decorators, and
feature converters that can be stacked the same way."""
import abc
import collections.abc

__all__ = ['FeatureConverter', 'MapFeatureConverter', 'Dataset', 'MapDataset', 'map_dataset']


class FeatureConverter(abc.ABC):
    @abc.abstractmethod
    def _train(self, dataarray, **kwargs):
        raise NotImplementedError

    def train(self, dataset, keys, **kwargs):
        from ..dataset import make_dataset_to_array
        self._train(make_dataset_to_array(dataset, keys), **kwargs)

    @abc.abstractmethod
    def convert(self, feature, **kwargs):
        raise NotImplementedError


class MapFeatureConverter(FeatureConverter):
    """A converter that pre/post-processes features around a base converter."""

    def __init__(self, base_converter):
        self.base = base_converter

    def _train(self, dataarray, **kwargs):
        return self.base._train(dataarray, **kwargs)

    def __getattr__(self, name):
        return getattr(self.base, name)

    @abc.abstractmethod
    def convert(self, feature, raw=None, **kwargs):
        if raw is None:
            raw = feature
        if isinstance(self.base, MapFeatureConverter):
            return self.base.convert(feature, raw, **kwargs)
        return self.base.convert(feature, **kwargs)


class Dataset(collections.abc.Mapping):
    @abc.abstractmethod
    def keys(self):
        raise NotImplementedError

    @abc.abstractmethod
    def get_data(self, key):
        raise NotImplementedError

    def __getitem__(self, key):
        return self.get_data(key)

    def __iter__(self):
        return ((key, self[key]) for key in self.keys())

    def __len__(self):
        return len(self.keys())


class MapDataset(Dataset):
    """Applies `function` to every item of a base dataset, on access."""
    expand_tuple = True   # apply to each member of a tuple item separately
    with_key = False      # pass key=...
    with_raw = False      # pass raw=<the undecorated item>

    def __init__(self, base_dataset, **kwargs):
        super().__init__()
        self.base = base_dataset
        self.kwargs = kwargs

    def keys(self):
        return self.base.keys()

    def __getattr__(self, name):
        return getattr(self.base, name)

    def get_data(self, key, with_raw=False):
        if isinstance(self.base, MapDataset):
            data, raw = self.base.get_data(key, with_raw=True)
        else:
            data = raw = self.base[key]

        extra = dict(self.kwargs)
        if self.with_key:
            extra['key'] = key

        if self.expand_tuple and isinstance(data, tuple):
            if self.with_raw:
                result = tuple(self.function(d, raw=r, **extra) for d, r in zip(data, raw))
            else:
                result = tuple(self.function(d, **extra) for d in data)
        else:
            if self.with_raw:
                extra['raw'] = raw
            result = self.function(data, **extra)

        return (result, raw) if with_raw else result

    @staticmethod
    @abc.abstractmethod
    def function(data):
        raise NotImplementedError


def map_dataset(expand_tuple=True, with_key=False, with_raw=False):
    """Decorator: turn a plain function into a MapDataset subclass."""
    def build(func):
        return type(func.__name__, (MapDataset,), {
            '__module__': func.__module__, '__doc__': func.__doc__,
            'function': staticmethod(func),
            'expand_tuple': expand_tuple, 'with_key': with_key, 'with_raw': with_raw,
        })
    return build
