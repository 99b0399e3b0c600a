"""Captures PairBatchPipeline steps in their variants (plain / wav in / pcm out / one or two waves / serial), one fresh
process per case, and reports which survive hipStreamEndCapture.  Round 5: with the f0 kernels at the head of every wave's
own stream (side streams forked at two depths of the graph) every two-wave wav-in case crashed inside the runtime (ROCm 7.2);
with the f0 stage of all waves on the origin stream before ONE fork all cases pass."""
import subprocess
import sys

CASES = ['plain', 'wav', 'pcm', 'wav_pcm', 'wav_pcm_norng', 'wav_pcm_head', 'wav_pcm_serial', 'wav_1wave', 'wav_pcm_1pair']
if len(sys.argv) == 1:
    for c in CASES:
        r = subprocess.run([sys.executable, '-X', 'faulthandler', __file__, c], capture_output=True, text=True)
        print(c, 'rc', r.returncode, r.stdout.strip()[-200:], ('\n' + r.stderr.strip()[-900:]) if r.returncode else '', flush=True)
    sys.exit(0)
case = sys.argv[1]
import numpy as np
import torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from kwiiyatta_amd import pipeline as pl
from kwiiyatta_amd.backend.nprandom import DeviceRandomState
from kwiiyatta_amd.synthetic import make_utterance
fs = 48000
npairs = 1 if case.endswith('1pair') else 4
pairs = [(make_utterance(seed=2 * i, fs=fs, seconds=1.0), make_utterance(seed=2 * i + 1, fs=fs, seconds=1.1)) for i in range(npairs)]
gmm = pl.synthetic_gmm(order=24, components=4, seed=0, n_frames=3000)
dg = pl.DeviceGMM(gmm.weights_, gmm.means_, gmm.covariances_, torch.device('cuda', 0))
kw = dict(wav_in='wav' in case, pcm='pcm' in case)
rng = None if 'norng' in case else DeviceRandomState.from_seed(3, device_index=0)
if 'head' in case:
    kw['rng_place'] = 'head'
if 'serial' in case:
    kw['serial'] = True
waves = 1 if ('1wave' in case or 'serial' in case) else 2
p = pl.PairBatchPipeline(0, fs, pairs, dg, waves=waves, rng=rng, **kw)
p.run(); p.sync(); torch.cuda.synchronize()
a = p.wave(0).clone()
p.capture()
p.replay(); p.sync(); torch.cuda.synchronize()
print('captured and replayed', case, 'finite', bool(torch.isfinite(p.wave(0)).all()), 'same', bool(torch.equal(a, p.wave(0))) if rng is None else '-')
