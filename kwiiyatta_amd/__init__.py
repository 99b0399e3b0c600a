"""kwiiyatta_amd -- MI355X-native implementation of kwiiyatta's per-utterance
conversion hot path (WORLD analysis, mel-cepstrum, FastDTW alignment, GMM/MLPG
conversion, WORLD synthesis) behind the reference's own Python API.

The numerics run in hand-written gfx950 HIP kernels (``libkwy.so``, C ABI in
``include/kwy.h``); there is no CPU fallback.
"""
name = "kwiiyatta_amd"
