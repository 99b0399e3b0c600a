import sys, numpy as np, torch
sys.path.insert(0,'.')
from kwiiyatta_amd.backend import dtw
from kwiiyatta_amd import _lib
from kwiiyatta_amd._lib import lib, c_vp
rng=np.random.default_rng(0)
def series(T, dim, warp):
    t=np.linspace(0,1,T)**warp
    base=np.stack([np.sin(2*np.pi*(k+1)*t*3+k) for k in range(dim)],1)
    return base+0.05*rng.standard_normal((T,dim))
x=series(2201,26,1.0); y=series(2401,26,1.3)
ctx=_lib.default_context()
dtw.fastdtw(x,y,radius=32)
dbg=torch.zeros(256,dtype=torch.int64,device='cuda')
lib.kwy_ctx_debug_buffer.argtypes = [c_vp, c_vp]   # diagnostic hook, not in include/kwy.h
lib.kwy_ctx_debug_buffer(ctx.handle, c_vp(dbg.data_ptr()))
dtw.fastdtw(x,y,radius=32)
d=dbg.cpu().numpy()
print('total dp cycles', d[0], 'total bt cycles', d[1], 'finest level dp', d[2], 'bt', d[3], 'path', d[4])
print('finest level trace: staged', d[20], 'hopped', d[21], 'planes staged', d[24], 'walked', d[22], 'counted', d[23], 'end', d[3])
print('trace flags (1 planes staged, 2 tables staged)', d[25], 'walks from memory', d[26], 'groups staged', d[27])
print('simd of the four wavefronts', [(int(v) >> 4) & 3 for v in d[16:20]], 'cu', [(int(v) >> 8) & 15 for v in d[16:20]])
print('finest level strips: k, start, end, steps, jmin')
for k in range(48):
    if d[64+4*k+1]: print(k, d[64+4*k], d[64+4*k+1], d[64+4*k+2], d[64+4*k+3])
