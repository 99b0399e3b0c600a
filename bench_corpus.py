#!/usr/bin/env python
"""BASELINE config 5: an atr503-sized parallel corpus end to end on the GPU(s) --
per-pair analysis + alignment -> joint training matrix kept in HBM -> converter fit (k-means
initialisation + EM, statistics all-reduced over RCCL) -> batch conversion of the source utterances.

    python bench_corpus.py [--pairs 503] [--seconds 5] [--components 64] [--em-iters 10]
    python -m torch.distributed.run --nproc-per-node G ... bench_corpus.py --gpus G

Pairs are sharded in contiguous blocks over the ranks (no collective before the fit); the fit's only
exchange is the all-reduce of its sufficient statistics; conversion shards the source utterances again.
Synthetic 48 kHz utterances (kwiiyatta_amd.synthetic, `--distinct` different pairs cycled to the corpus
size -- generating 1006 distinct signals on the host would take longer than the whole run), f0 tracks given.
Prints ONE JSON line: frames/s per phase and the wall time of each (strong scaling: the corpus is fixed).
The silence padding of align_even comes from numpy's legacy generator as in the reference: reproduced on the GPU by
default (--pads device: one stream of draws for the whole corpus, every rank skips to its block, so the training
matrix is the same for any number of ranks), or drawn by np.random on the host (--pads host; its time is inside the
data-set phase and reported separately as well).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
# one hardware queue per stream and a few to spare (bench.py has the measurements); read once when HIP starts
os.environ.setdefault('GPU_MAX_HW_QUEUES', '64')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--pairs', type=int, default=503)
    ap.add_argument('--seconds', type=float, default=5.0)
    ap.add_argument('--distinct', type=int, default=8)
    ap.add_argument('--components', type=int, default=64)
    ap.add_argument('--em-iters', type=int, default=10, help='EM iterations (tol=0: exactly this many)')
    ap.add_argument('--streams', type=int, default=16)
    ap.add_argument('--convert', type=int, default=None, help='source utterances to convert (default: all)')
    ap.add_argument('--max-iter', type=int, default=None,
                    help='fit as the reference does: GaussianMixture(max_iter=N, tol=1e-3), stopping when converged '
                         '(kwiiyatta/converter/gmm.py:14-26 uses 100); default: exactly --em-iters iterations (tol=0)')
    ap.add_argument('--pads', choices=['device', 'host'], default='device',
                    help='pad spectra of align_even: numpy\'s legacy stream reproduced on the GPU (every rank advances '
                         'it past the pairs of the ranks before it: the training matrix does not depend on the number '
                         'of ranks), or np.random on the host as the reference draws them (then seeded per rank)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL; gloo lets '
                                                       'several ranks share one GPU when rehearsing the launch)')
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = local_rank % max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.backend)
    torch.cuda.set_device(local_rank)
    from kwiiyatta_amd import corpus as cp
    from kwiiyatta_amd.backend.nprandom import DeviceRandomState
    from kwiiyatta_amd.converter.gmm_fit import GaussianMixtureHIP
    from kwiiyatta_amd.parallel import gather_frame_counts
    from kwiiyatta_amd.synthetic import make_utterance
    fs = 48000
    distinct = []
    for k in range(args.distinct):
        distinct.append((make_utterance(seed=1000 + k, fs=fs, seconds=args.seconds),
                         make_utterance(seed=5000 + k, fs=fs, seconds=args.seconds, time_warp=1.1, formant_scale=1.12)))
    mine = cp.shard_block(args.pairs, rank, world)
    pairs = [distinct[i % args.distinct] for i in mine]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # one pool of streams and library contexts for all phases (every extra live stream is one more hardware queue)
    pool = cp.StreamPool(local_rank, args.streams)

    # warm-up: tables, arenas, RCCL channels
    np.random.seed(rank)
    cp.build_training_matrix(pairs[:2], fs, device_index=local_rank, pool=pool)
    barrier()

    # ---- phase 1: data set -------------------------------------------------------------------------
    # the pad spectra come from numpy's global generator in pair order (the reference's semantics): drawn by a helper
    # thread one wave of pairs ahead of the GPU; the generator alone is timed separately on a small sample
    np.random.seed(4321 + rank)
    K = 1025
    t0 = time.perf_counter()
    for _ in range(8):
        [cp.draw_silence(fs, K) for _ in range(4)]
    t_rng = (time.perf_counter() - t0) / 8 * len(pairs)
    np.random.seed(1234 + rank)
    t0 = time.perf_counter()
    if args.pads == 'device':
        # one stream of draws for the whole corpus, seeded like np.random.seed(1234): this rank skips the pairs before
        # its block (inside the timed region) and gets exactly the pads a one-rank run gives those pairs
        rng = DeviceRandomState.from_seed(1234, device_index=local_rank)
        X, frames = cp.build_training_matrix(pairs, fs, device_index=local_rank, pool=pool, rng=rng,
                                             pairs_before=mine[0] if mine else 0)
    else:
        X, frames = cp.build_training_matrix(pairs, fs, device_index=local_rank, pool=pool)
    barrier()
    t_data = time.perf_counter() - t0

    # ---- phase 2: fit --------------------------------------------------------------------------------
    # (one-time set-up of the fit -- kernel attributes, RCCL channels of the statistics' all-reduce -- on a few rows,
    # as the other two phases are warmed up; the buffers of the timed fit are allocated inside it)
    GaussianMixtureHIP(n_components=args.components, max_iter=1, tol=0.0, random_state=0,
                       device_index=local_rank).fit(X[:max(4 * args.components, 1024)])
    barrier()
    t0 = time.perf_counter()
    fit_opts = dict(max_iter=args.em_iters, tol=0.0) if args.max_iter is None else dict(max_iter=args.max_iter, tol=1e-3)
    g = GaussianMixtureHIP(n_components=args.components, random_state=0, device_index=local_rank, **fit_opts).fit(X)
    barrier()
    t_fit = time.perf_counter() - t0
    rows_local = X.shape[0]
    # a fingerprint of the fitted model and of the training matrix: equal for every number of ranks (--pads device)
    x_sum = torch.tensor([float(X.sum().item()), float(X.shape[0])], dtype=torch.float64,
                         device=X.device if args.backend == 'nccl' else 'cpu')
    if world > 1:
        dist.all_reduce(x_sum)
    fingerprint = {'training_matrix_sum': float(x_sum[0].item()), 'rows': int(x_sum[1].item()),
                   'means_sum': float(np.sum(g.means_)), 'lower_bound': float(g.lower_bound_)}
    del X

    # ---- phase 3: batch conversion ---------------------------------------------------------------------
    n_conv = args.pairs if args.convert is None else args.convert
    conv_idx = cp.shard_block(n_conv, rank, world)
    sources = [distinct[i % args.distinct][0] for i in conv_idx]
    cp.convert_batch(sources[:2], fs, g, device_index=local_rank, pool=pool)      # warm-up
    barrier()
    t0 = time.perf_counter()
    waves = cp.convert_batch(sources, fs, g, device_index=local_rank, pool=pool)
    barrier()
    t_conv = time.perf_counter() - t0
    conv_frames = sum(len(s[1]) for s in sources)
    del waves

    tot_frames, t_data_max = gather_frame_counts(frames, t_data)
    tot_rows, t_fit_max = gather_frame_counts(rows_local, t_fit)
    tot_conv, t_conv_max = gather_frame_counts(conv_frames, t_conv)
    _, t_rng_max = gather_frame_counts(0, t_rng)
    if rank == 0:
        total = t_data_max + t_fit_max + t_conv_max
        print(json.dumps({
            'metric': 'frames/sec, corpus: analyse+align -> fit -> batch conversion, 48 kHz 5 ms hop',
            'value': (tot_frames + tot_conv) / total, 'unit': 'frames/s', 'n_gpus': world,
            'higher_is_better': True, 'scaling': 'strong', 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'config5: {args.pairs} parallel pairs x {args.seconds:g} s (48 kHz), '
                                   f'{args.distinct} distinct pairs cycled; align_even(pad 100, radius 32, strict), '
                                   f'joint static+delta+delta2 rows (D=144) in HBM; GMM {args.components} full-covariance '
                                   f'components, k-means init + {args.em_iters} EM iterations; MLPG conversion + synthesis '
                                   f'of {n_conv} source utterances',
                       'warmup': 'each phase once on 2 pairs / 1024 rows / 2 utterances before its timed run (tables, kernel set-up, RCCL channels)',
                       'parallelism': f'pairs in contiguous blocks over {world} rank(s), {args.streams} streams each; '
                                      f'all-reduce of the fit statistics only'},
            'phases': {
                'dataset': {'seconds': t_data_max, 'source_frames': tot_frames, 'frames_per_s': tot_frames / t_data_max,
                            'joint_rows': tot_rows,
                            'host_rng_seconds': t_rng_max if args.pads == 'host' else 0.0},
                'fit': {'seconds': t_fit_max, 'kmeans_lloyd_iterations': g.kmeans_n_iter_, 'em_iterations': g.n_iter_,
                        'converged': bool(g.converged_),
                        'stopping': 'tol=0: exactly --em-iters iterations' if args.max_iter is None else
                                    f'max_iter={args.max_iter}, tol=1e-3 (the reference\'s GaussianMixture settings)',
                        'rows_per_s': tot_rows * g.n_iter_ / t_fit_max},
                'convert': {'seconds': t_conv_max, 'frames': tot_conv, 'frames_per_s': tot_conv / t_conv_max},
            },
            'pads': args.pads, 'backend': args.backend if world > 1 else None, 'fingerprint': fingerprint,
            'total_seconds': total}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
