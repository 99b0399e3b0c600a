"""Drop-in replacements for the third-party numerics the reference calls.

Each module mirrors the call signatures of one upstream package, backed by the
HIP kernels in ``libkwy.so`` (no CPU fallback):

  ``world``  -> pyworld   (reference: kwiiyatta/vocoder/world.py:35-96)
  ``sptk``   -> pysptk    (reference: kwiiyatta/vocoder/mcep.py:26,65,71)
  ``dtw``    -> fastdtw   (reference: kwiiyatta/vocoder/align.py:71)
  ``mlpg``   -> nnmnkwii  (reference: kwiiyatta/converter/delta.py, gmm.py)
"""
