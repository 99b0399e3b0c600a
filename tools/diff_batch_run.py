"""The batch path's two outputs per file -- .synth.wav and .diff.wav (the input through the MLSA filter of the
differential conversion) -- for 64 synthetic 48 kHz utterances of 5 s, wav in -> 16-bit PCM out on the device
(corpus.convert_batch(pcm=True, diff=True): what `kwiiyatta --batch` runs).  Under rocprofv3 --stats this gives the
k_mlsa_filter / k_mc2b / k_fin_* rows of profiles/; prints one JSON line with the wall times with and without the
differential outputs."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kwiiyatta_amd import corpus, pipeline as pl          # noqa: E402
from kwiiyatta_amd.synthetic import make_utterance         # noqa: E402

fs, n, sec = 48000, 64, 5.0
waves = [make_utterance(seed=900 + i, fs=fs, seconds=sec, f0_base=100.0 + 7 * (i % 9))[0] for i in range(n)]
gmm = pl.synthetic_gmm(order=24, components=64, seed=0)
ls = corpus._Lockstep(0)
dev = [torch.from_numpy(w).cuda() for w in waves]
res = {}
for diff in (False, True):
    corpus.convert_batch(dev[:16], fs, gmm, pcm=True, diff=diff, lockstep=ls)        # warm-up: tables, arenas
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = corpus.convert_batch(dev, fs, gmm, pcm=True, diff=diff, lockstep=ls)
    torch.cuda.synchronize()
    res['with_diff' if diff else 'synth_only'] = time.perf_counter() - t0
frames = sum(int(fs * sec) * 1000 // fs // 5 + 1 for _ in waves)
print(json.dumps({'workload': f'{n} x {sec:g} s 48 kHz utterances, wav in -> int16 out, GMM 64 components',
                  'frames': frames, 'seconds_synth_only': res['synth_only'], 'seconds_with_diff_outputs': res['with_diff'],
                  'frames_per_s_with_diff': frames / res['with_diff'],
                  'diff_pcm_peak': int(max(int(p.abs().max().item()) for p in out[3]))}))
