"""In-kernel stamps of the synthesis' phase chain (k_syn_phase) on the benchmark's target utterance."""
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
dbg = torch.zeros(256, dtype=torch.int64, device='cuda')
lib.kwy_ctx_debug_buffer.argtypes = [c_vp, c_vp]   # diagnostic hook, not in include/kwy.h
lib.kwy_ctx_debug_buffer(ctx.handle, c_vp(dbg.data_ptr()))
world.synthesize(f0, sp, ap, fs)
torch.cuda.synchronize()
d = dbg.cpu().numpy()
print('k_syn_phase: %d clock64 ticks = %.1f us on the 100 MHz wall clock (%.2f ticks per ns)' % (d[40], d[41] / 100.0, d[40] / max(d[41] * 10.0, 1)))
print('fast tiles %d: %d ticks; slow tiles %d: %d ticks in %d rounds' % (d[44], d[42], d[45], d[43], d[46]))
