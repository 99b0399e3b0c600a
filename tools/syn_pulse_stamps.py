"""In-kernel stamps of one pulse of k_syn_pulse (a workgroup in the middle of the grid, its first voiced pulse)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from kwiiyatta_amd import _lib
from kwiiyatta_amd._lib import lib, c_vp
from kwiiyatta_amd.backend import world
from kwiiyatta_amd.synthetic import make_utterance

fs = 48000
x, f0, t = make_utterance(seed=1, fs=fs, seconds=11.0)
f0, t = world.dio(x, fs)
sp = world.cheaptrick(x, f0, t, fs)
ap = world.d4c(x, f0, t, fs)
ctx = _lib.default_context()
world.synthesize(f0, sp, ap, fs)
dbg = torch.zeros(64, dtype=torch.int64, device='cuda')
dbg[63] = 700
lib.kwy_ctx_debug_buffer.argtypes = [c_vp, c_vp]   # diagnostic hook, not in include/kwy.h
lib.kwy_ctx_debug_buffer(ctx.handle, c_vp(dbg.data_ptr()))
world.synthesize(f0, sp, ap, fs)
torch.cuda.synchronize()
d = dbg.cpu().numpy()
names = {1: 'rows -> log spectra', 2: 'periodic: 2 rfft + exp', 3: 'shift', 4: 'irfft', 5: 'dc sum',
         7: 'aperiodic: swap, 3 rfft + exp + noise', 8: 'product', 9: 'irfft', 10: '-', 11: 'output'}
last = d[0]
for i in sorted(names):
    print(names[i].ljust(40), d[i] - last)
    last = d[i]
print('total', d[11] - d[0])
