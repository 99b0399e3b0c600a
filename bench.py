#!/usr/bin/env python
"""Benchmark of the kwiiyatta hot path on MI355X.

`python bench.py --gpus N --steps K --warmup W` (N>1: launched under
torch.distributed.run, one rank per GPU).  A step = one pass of the hot path
over one batch of synthetic 48 kHz / 10 s utterances that are already resident
in HBM; utterances are independent, so ranks shard them with no collective
(weak scaling: per-GPU batch fixed).  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 48000
SECONDS = 10.0
FRAME_PERIOD = 5.0


def cpu_baseline(x, f0, t, seconds_budget=20.0):
    """The CPU oracle ("port" of the reference's pyworld path) timed on one host
    core over a bounded sample of the same workload."""
    from oracle import oracle as ko
    frames = int(200 * 2.0) + 1          # first 2 s of the utterance
    n = int(FS * 2.0)
    xs, f0s, ts = np.ascontiguousarray(x[:n]), np.ascontiguousarray(f0[:frames]), np.ascontiguousarray(t[:frames])
    reps, t0 = 0, time.perf_counter()
    while True:
        sp = ko.cheaptrick(xs, f0s, ts, FS)
        ap = ko.d4c(xs, f0s, ts, FS)
        ko.synthesize(f0s, sp, ap, FS, FRAME_PERIOD)
        reps += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or reps >= 8:
            break
    return {'value': reps * frames / el, 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
            'sample': f'{reps}x first 2 s (401 frames) of the utterance: cheaptrick+d4c+synthesize, '
                      f'oracle/liboracle.so (C restatement of pyworld 0.2.8), 1 thread'}


def main():
    ap_ = argparse.ArgumentParser()
    ap_.add_argument('--gpus', type=int, default=1)
    ap_.add_argument('--steps', type=int, default=20)
    ap_.add_argument('--warmup', type=int, default=3)
    ap_.add_argument('--batch', type=int, default=8, help='utterances per GPU per step')
    ap_.add_argument('--no-cpu-baseline', action='store_true')
    args = ap_.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    from kwiiyatta_amd import _lib
    from kwiiyatta_amd._lib import lib, c_vp
    from kwiiyatta_amd.synthetic import make_utterance

    # distinct utterances per rank (seed = global utterance index); a few base
    # signals are generated on the host and reused round-robin inside the batch
    nbase = min(args.batch, 2)
    base = [make_utterance(seed=1234 + rank * args.batch + i, fs=FS, seconds=SECONDS) for i in range(nbase)]
    T = len(base[0][1])
    N = len(base[0][0])
    fft = lib.kwy_cheaptrick_fft_size(FS, 71.0)
    K = fft // 2 + 1
    ylen = lib.kwy_synth_length(T, FRAME_PERIOD, FS)

    streams = [torch.cuda.Stream(device=dev) for _ in range(args.batch)]
    ctxs = [_lib.Context(local_rank, stream=s.cuda_stream) for s in streams]
    bufs = []
    for i in range(args.batch):
        x, f0, t = base[i % nbase]
        bufs.append(dict(
            x=torch.from_numpy(x).to(dev), f0=torch.from_numpy(f0).to(dev), t=torch.from_numpy(t).to(dev),
            sp=torch.empty((T, K), dtype=torch.float64, device=dev),
            ap=torch.empty((T, K), dtype=torch.float64, device=dev),
            y=torch.empty(ylen, dtype=torch.float64, device=dev)))
    torch.cuda.synchronize()

    def p(tensor):
        return c_vp(tensor.data_ptr())

    def step():
        for ctx, b in zip(ctxs, bufs):
            h = ctx.handle
            _lib.check(ctx, lib.kwy_cheaptrick_dev(h, p(b['x']), N, FS, p(b['t']), p(b['f0']), T, -0.15, 71.0, fft, float(FS), p(b['sp'])))
            _lib.check(ctx, lib.kwy_d4c_dev(h, p(b['x']), N, FS, p(b['t']), p(b['f0']), T, 0.85, fft, p(b['ap'])))
            _lib.check(ctx, lib.kwy_synthesize_dev(h, p(b['f0']), T, p(b['sp']), p(b['ap']), fft, FRAME_PERIOD, FS, float(FS), ylen, p(b['y'])))

    def sync_all():
        for ctx in ctxs:
            ctx.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    if world > 1:
        dist.barrier()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())

    frames_total = world * args.batch * T * args.steps
    value = frames_total / el

    if rank == 0:
        # per-frame algorithmic bytes of the timed path: analyse 18 336 + synth 18 328
        bytes_per_frame = (240 * 8 + 16 + 2 * K * 8) + (2 * K * 8 + 8 + 240 * 8)
        out = {
            'metric': 'frames/sec end-to-end analyse->align->convert->synth, 48 kHz 5 ms hop',
            'value': value, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1000.0 * el / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'config2: 48 kHz 10 s utterances (T=2001, K=1025): CheapTrick + D4C + WORLD synthesis '
                                   '(align/convert stages not in this round-1 line yet)',
                       'utterances_per_gpu': args.batch, 'frames_per_utterance': T,
                       'streams_per_gpu': args.batch, 'parallelism': f'utterance-sharded x{world}'},
            'real_time_factor': value / 200.0,
            'hbm_fraction_whole_path': value / world * bytes_per_frame / 8e12,
            'roofline': None,
            'cpu_baseline': None,
        }
        if not args.no_cpu_baseline and world == 1:
            x, f0, t = base[0]
            out['cpu_baseline'] = cpu_baseline(x, f0, t)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
