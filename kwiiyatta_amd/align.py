"""`kwiiyatta.align` dispatcher: features are aligned directly, datasets become
an aligned parallel dataset (mirrors /root/reference/kwiiyatta/align.py:7-19)."""
import kwiiyatta_amd as kwiiyatta
from . import converter, vocoder


def align(a, b, **kwargs):
    for base, handler in ((vocoder.abc.Feature, lambda: vocoder.align(a, b, **kwargs)),
                          (converter.abc.Dataset,
                           lambda: kwiiyatta.align_dataset(kwiiyatta.ParallelDataset(a, b)))):
        if isinstance(a, base):
            if not isinstance(b, base):
                raise TypeError(f'argument type mismatch: {type(a)!r} and {type(b)!r}')
            return handler()
    raise TypeError('argument should be Feature or Dataset')
