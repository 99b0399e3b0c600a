"""`kwiiyatta.align(a, b)`: two feature sets are aligned directly (vocoder.align), two datasets become the aligned
parallel dataset of their common keys (/root/reference/kwiiyatta/align.py)."""
from . import converter, vocoder


def align(a, b, **kwargs):
    if isinstance(a, vocoder.abc.Feature):
        kind, run = vocoder.abc.Feature, lambda: vocoder.align(a, b, **kwargs)
    elif isinstance(a, converter.abc.Dataset):
        kind, run = converter.abc.Dataset, lambda: converter.align_dataset(converter.ParallelDataset(a, b))
    else:
        raise TypeError('argument should be Feature or Dataset')
    if not isinstance(b, kind):
        raise TypeError(f'argument type mismatch: {type(a)!r} and {type(b)!r}')
    return run()
