"""Dynamic-feature stage of the converter stack: static mel-cepstra become [static | delta | delta-delta] for
training, and conversion results are cut back to the static part.  API of kwiiyatta.converter.delta
(/root/reference/kwiiyatta/converter/delta.py).  The windows are the reference's; `delta_features` is nnmnkwii's
function restated in kwiiyatta_amd.backend.mlpg (the device-resident paths use the kwy_delta_features kernel)."""
from ..backend.mlpg import DELTA_WINDOWS, delta_features
from . import abc

__all__ = ['DELTA_WINDOWS', 'DeltaFeatureDataset', 'DeltaFeatureConverter']


class DeltaFeatureDataset(abc.MapDataset):
    """delta features of every item; all items must share one frame period (the deltas are per frame)"""
    with_key = True
    with_raw = True

    def __init__(self, base):
        super().__init__(base)
        self.frame_period = None

    def function(self, feature, raw, key):
        period = raw.frame_period
        if self.frame_period is None:
            self.frame_period = period
        if period != self.frame_period:
            raise ValueError(f'frame_period of "{key}" is {period!r} but others are {self.frame_period!r}')
        return delta_features(feature, DELTA_WINDOWS)


class DeltaFeatureConverter(abc.MapFeatureConverter):
    def train(self, dataset, keys, **kwargs):
        with_deltas = DeltaFeatureDataset(dataset)
        self.base.train(with_deltas, keys, **kwargs)
        self.frame_period = with_deltas.frame_period

    def convert(self, feature, raw, **kwargs):
        if raw.frame_period != self.frame_period:
            raise ValueError(f'frame_period is expected to {self.frame_period!s} but {raw.frame_period!s}')
        static_dim = feature.shape[-1]
        converted = super().convert(delta_features(feature, DELTA_WINDOWS), **kwargs)
        return converted[:, :static_dim]
