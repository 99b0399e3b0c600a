"""Delta-feature stage of the converter stack (mirrors
/root/reference/kwiiyatta/converter/delta.py:8-50)."""
from ..backend.mlpg import DELTA_WINDOWS, delta_features
from . import abc

__all__ = ['DELTA_WINDOWS', 'DeltaFeatureDataset', 'DeltaFeatureConverter']


class DeltaFeatureDataset(abc.MapDataset):
    with_key = True
    with_raw = True

    def __init__(self, base):
        super().__init__(base)
        self.frame_period = None

    def function(self, feature, raw, key):
        if self.frame_period is None:
            self.frame_period = raw.frame_period
        elif self.frame_period != raw.frame_period:
            raise ValueError(f'frame_period of "{key}" is {raw.frame_period!r}'
                             f' but others are {self.frame_period!r}')
        return delta_features(feature, DELTA_WINDOWS)


class DeltaFeatureConverter(abc.MapFeatureConverter):
    def train(self, dataset, keys, **kwargs):
        delta_dataset = DeltaFeatureDataset(dataset)
        self.base.train(delta_dataset, keys, **kwargs)
        self.frame_period = delta_dataset.frame_period

    def convert(self, feature, raw, **kwargs):
        if self.frame_period != raw.frame_period:
            raise ValueError(f'frame_period is expected to {self.frame_period!s}'
                             f' but {raw.frame_period!s}')
        dim = feature.shape[-1]
        # the base converter recomputes the deltas on the GPU from the static part
        result = super().convert(delta_features(feature, DELTA_WINDOWS), **kwargs)
        if result.shape[-1] > dim:
            result = result[:, :dim]
        return result
