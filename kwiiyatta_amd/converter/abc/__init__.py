"""Composable building blocks of the converter layer.  API of kwiiyatta.converter.abc
(/root/reference/kwiiyatta/converter/abc/): `Dataset` (a read-only mapping key -> item, evaluated on access),
`MapDataset` / `map_dataset` (a dataset that applies a function to another dataset's items) and the matching pair
for converters, `FeatureConverter` / `MapFeatureConverter`.

Stacks are plain object chains: every wrapper holds its `base` and forwards unknown attributes to it, so the
outermost object answers for the whole stack (`stack.frame_period`, `stack.gmm`, ...).  A wrapped dataset hands
the *raw* item (what the innermost, non-mapping dataset produced) along with the processed one, because later
stages need its metadata (frame period, sampling rate) after the arrays have lost it."""
import abc
import collections.abc

__all__ = ['FeatureConverter', 'MapFeatureConverter', 'Dataset', 'MapDataset', 'map_dataset']


class _Forwarding:
    """attribute look-ups that fail here continue at `self.base`"""

    def __getattr__(self, name):
        if name == 'base':                       # not set yet (during construction / unpickling)
            raise AttributeError(name)
        return getattr(self.base, name)


# ---- datasets ----------------------------------------------------------------------------------------------------
class Dataset(collections.abc.Mapping):
    @abc.abstractmethod
    def keys(self):
        raise NotImplementedError

    @abc.abstractmethod
    def get_data(self, key):
        raise NotImplementedError

    def __getitem__(self, key):
        return self.get_data(key)

    def __len__(self):
        return len(self.keys())

    def __iter__(self):
        for key in self.keys():
            yield key, self[key]              # (key, item) pairs, as the reference iterates


class MapDataset(_Forwarding, Dataset):
    """`function` applied to the items of `base`.  Class switches: `expand_tuple` -- apply to every member of a
    tuple item (parallel data) instead of to the tuple; `with_key` / `with_raw` -- pass key= / raw= as well."""
    expand_tuple = True
    with_key = False
    with_raw = False

    def __init__(self, base_dataset, **kwargs):
        self.base = base_dataset
        self.kwargs = kwargs

    def keys(self):
        return self.base.keys()

    def _source(self, key):
        """(item to process, raw item)"""
        if isinstance(self.base, MapDataset):
            return self.base.get_data(key, with_raw=True)
        item = self.base[key]
        return item, item

    def get_data(self, key, with_raw=False):
        item, raw = self._source(key)
        options = dict(self.kwargs, key=key) if self.with_key else dict(self.kwargs)

        def apply(member, member_raw):
            if self.with_raw:
                return self.function(member, raw=member_raw, **options)
            return self.function(member, **options)

        if self.expand_tuple and isinstance(item, tuple):
            out = tuple(apply(m, r) for m, r in zip(item, raw))
        else:
            out = apply(item, raw)
        return (out, raw) if with_raw else out

    @staticmethod
    @abc.abstractmethod
    def function(data):
        raise NotImplementedError


def map_dataset(expand_tuple=True, with_key=False, with_raw=False):
    """decorator: `@map_dataset() def Name(item, **kw)` -> a MapDataset subclass called Name"""
    switches = dict(expand_tuple=expand_tuple, with_key=with_key, with_raw=with_raw)

    def subclass(func):
        body = dict(switches, function=staticmethod(func), __module__=func.__module__, __doc__=func.__doc__)
        return type(func.__name__, (MapDataset,), body)
    return subclass


# ---- converters ------------------------------------------------------------------------------------------------------
class FeatureConverter(abc.ABC):
    """train(dataset, keys) stacks the dataset's items into one matrix and hands it to `_train`"""

    @abc.abstractmethod
    def _train(self, dataarray, **kwargs):
        raise NotImplementedError

    @abc.abstractmethod
    def convert(self, feature, **kwargs):
        raise NotImplementedError

    def train(self, dataset, keys, **kwargs):
        from ..dataset import make_dataset_to_array
        self._train(make_dataset_to_array(dataset, keys), **kwargs)


class MapFeatureConverter(_Forwarding, FeatureConverter):
    """a conversion stage around `base`; subclasses transform the feature and call `super().convert`, which passes
    the raw feature on only to stages that take one"""

    def __init__(self, base_converter):
        self.base = base_converter

    def _train(self, dataarray, **kwargs):
        return self.base._train(dataarray, **kwargs)

    @abc.abstractmethod
    def convert(self, feature, raw=None, **kwargs):
        if isinstance(self.base, MapFeatureConverter):
            return self.base.convert(feature, feature if raw is None else raw, **kwargs)
        return self.base.convert(feature, **kwargs)
