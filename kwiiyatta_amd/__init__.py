"""kwiiyatta_amd -- MI355X-native implementation of kwiiyatta's per-utterance
conversion hot path (WORLD analysis, mel-cepstrum, FastDTW alignment, GMM/MLPG
conversion, WORLD synthesis) behind the reference's own Python API
(same export list as /root/reference/kwiiyatta/__init__.py:1-22; the host-side
feature model, converter stack and CLIs are this package's own code written
to that API).

The numerics run in hand-written gfx950 HIP kernels (``libkwy.so``, C ABI in
``include/kwy.h``) reached through the pyworld / pysptk / fastdtw / nnmnkwii
shaped modules in ``kwiiyatta_amd.backend``; there is no CPU fallback.
"""
from . import wavfile
from .wavfile import Wavdata, load_wav
from .vocoder import (Analyzer, Feature, MelCepstrum, Synthesizer, align_even, analyze_wav,
                      feature, pad_silence, resample, reshape)
from .converter import MelCepstrumConverter, ParallelDataset, WavFileDataset, align_dataset
from .filter import apply_mlsa_filter
from .align import align
from .config import Config

name = "kwiiyatta_amd"

__all__ = ['align', 'Config', 'MelCepstrumConverter', 'ParallelDataset', 'WavFileDataset',
           'align_dataset', 'apply_mlsa_filter', 'Analyzer', 'Feature', 'MelCepstrum',
           'Synthesizer', 'align_even', 'analyze_wav', 'feature', 'pad_silence', 'resample',
           'reshape', 'Wavdata', 'load_wav']
