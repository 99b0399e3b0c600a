"""DTW status / trace flags of a wave of pairs, f0 given against f0 extracted (wav in)."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
import bench
from kwiiyatta_amd import _lib, pipeline as pl
from kwiiyatta_amd._lib import lib, c_vp
utts = [bench._make_utterance_job(j) for i in range(4) for j in bench.pair_jobs(i, 5.0)]
pairs = [(utts[2 * i], utts[2 * i + 1]) for i in range(4)]
gmm = pl.synthetic_gmm(order=24, components=8, seed=0)
dev = torch.device('cuda', 0)
dgmm = pl.DeviceGMM(gmm.weights_, gmm.means_, gmm.covariances_, dev)
lib.kwy_ctx_debug_buffer.argtypes = [c_vp, c_vp]
for wav in (False, True):
    p = pl.PairBatchPipeline(0, 48000, pairs, dgmm, waves=1, serial=True, wav_in=wav, pcm=wav)
    p.run(); p.sync()
    dbg = torch.zeros(256, dtype=torch.int64, device='cuda')
    lib.kwy_ctx_debug_buffer(p.ctx.handle, c_vp(dbg.data_ptr()))
    p.ctx.profile(True) if hasattr(p.ctx, 'profile') else None
    p.run(); p.sync()
    d = dbg.cpu().numpy()
    print('wav_in', wav, 'bt cycles total', d[1], 'last trace: flags', d[25], 'walks from memory', d[26], 'groups', d[27], 'cycles', d[3], 'path', d[4],
          'stamps staged/hop/planes/walked/counted', d[20], d[21], d[24], d[22], d[23])
    for k in range(4):
        path, plen, dist = p.path(k)
        wv = p.waves[0]
        print('  pair', k, 'path_len', int(plen.item()), 'dist', float(dist.item()), 'T', wv.T[2 * k], wv.T[2 * k + 1])
    lib.kwy_ctx_debug_buffer(p.ctx.handle, c_vp(0))
