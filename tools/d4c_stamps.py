import sys, numpy as np, torch
sys.path.insert(0,'.')
from kwiiyatta_amd import _lib, pipeline as pl
from kwiiyatta_amd._lib import lib, c_vp
from kwiiyatta_amd.synthetic import make_utterance
u = make_utterance(seed=1234, fs=48000, seconds=10.0)
p = pl.UtterancePipeline(0, 48000, u)
dbg = torch.zeros(64, dtype=torch.int64, device='cuda')
f0 = u[1]
frame = int(np.where(f0>0)[0][200]); dbg[63] = frame
lib.kwy_ctx_debug_buffer.argtypes = [c_vp, c_vp]   # diagnostic hook, not in include/kwy.h
lib.kwy_ctx_debug_buffer(p.ctx.handle, c_vp(dbg.data_ptr()))
p.run(); p.sync(); p.run(); p.sync()
d = dbg.cpu().numpy()
names = ['start','rng','win0','fft+cen0','win1','fft+cen1','dccorr','win2','fft2','pow+smooth','gd smooth x2']
print('frame', frame, 'f0', f0[frame])
for i in range(1,11):
    print(names[i].ljust(16), d[i]-d[i-1])
print('body total', d[15]-d[0])
names2 = ['bands start','band1 head','band1 fft','band1 bins','band1 select','bands loop end','output']
for i in range(17,23):
    print(names2[i-16].ljust(16), d[i]-d[i-1])
print('bands total', d[22]-d[16])
print('select ctl words (band 0/2/4, band 1/3):', d[24:36], d[36:48])
