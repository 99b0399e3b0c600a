#!/usr/bin/env python
"""Config 5 companion benchmark: the converter fit (EM) on an atr503-sized joint
feature matrix (n ~ 5e5 frames x 144, 64 full-covariance components).

    python bench_fit.py [--frames N] [--iters K]                      one GPU
    python bench_fit.py --gpus G                                      G GPUs (G ranks started by bench_launch.py, or
                                                                     run under torch.distributed.run): frames
                                                                     sharded, statistics all-reduced (RCCL)
Prints one JSON line with the time per EM iteration (strong scaling: total frames fixed)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1, help='N > 1 outside a rank: start N ranks, one per GPU')
    ap.add_argument('--frames', type=int, default=500000)
    ap.add_argument('--dim', type=int, default=144)
    ap.add_argument('--components', type=int, default=64)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL; gloo lets '
                                                       'several ranks share one GPU when rehearsing the launch)')
    ap.add_argument('--force-group', action='store_true',
                    help='initialise the process group even for ONE rank (a world-size-1 \'nccl\' group: RCCL itself under '
                         'the fit\'s collectives on a one-GPU box)')
    args = ap.parse_args()
    import bench_launch
    bench_launch.maybe_launch(args.gpus)     # --gpus N > 1 outside a rank: become the launcher of N ranks (never returns)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = local_rank % max(1, torch.cuda.device_count())
    grouped = world > 1 or args.force_group
    if grouped:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29549')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.backend)
    torch.cuda.set_device(local_rank)
    from kwiiyatta_amd.converter.gmm_fit import Comm, HipStats, kmeans_init
    n_local = args.frames // world
    rng = np.random.default_rng(1000 + rank)
    M, D = args.components, args.dim
    centres = np.random.default_rng(0).standard_normal((M, D)) * 2
    lab = rng.integers(0, M, n_local)
    X = centres[lab] + rng.standard_normal((n_local, D))
    st = HipStats(X, M, device_index=local_rank)
    comm = Comm()

    def barrier():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()

    torch.cuda.set_stream(st.stream)           # the fit's stream: kernels, driver ops and collectives
    # initialisation: k-means++ and Lloyd on the GPUs (inside the measurement, reported separately)
    kmeans_init(st, M, 0, max_iter=2)          # warm-up: tables, arena, RCCL channels
    barrier()
    t0 = time.perf_counter()
    km_iters, _ = kmeans_init(st, M, 0)
    barrier()
    km_s = time.perf_counter() - t0

    def m_step():
        s = comm.all_reduce(st.sums())
        st.means_from(s)
        c = comm.all_reduce(st.cov(s))
        st.finalize(s, c, 1e-6)

    def e_step():
        both = comm.all_reduce(torch.cat((st.estep(), st.estep_failed())))
        return both.tolist()                   # the host reads the log-likelihood every iteration, as the fit does

    m_step()
    e_step(); m_step()        # warm-up iteration
    barrier()
    st.ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.iters):
        e_step()
        m_step()
    barrier()
    el = time.perf_counter() - t0
    ranks_seen = int(round(float(comm.all_reduce(torch.ones(1, dtype=torch.float64, device=st.X.device)).item()))) if grouped else 1
    if rank == 0:
        lp, n1 = st.ctx.profile_read('k_fit_logprob')
        cv, n2 = st.ctx.profile_read('k_fit_cov')
        flops = 2.0 * n_local * M * (D * (D + 1) / 2)          # triangular products: Z x per frame, lower half of sum r d d'
        lp_ms, cv_ms = lp / max(n1, 1), cv / max(n2, 1)
        peak = 78.6                                            # v_mfma_f64_16x16x4_f64: 64 cycles (tools/mfma_f64_bench.hip)
        lp_tf = flops / (lp_ms * 1e-3) / 1e12 if n1 else None
        cv_tf = flops / (cv_ms * 1e-3) / 1e12 if n2 else None
        print(json.dumps({'metric': 'EM iteration time, full-covariance GMM fit', 'value': el / args.iters * 1e3,
                          'unit': 'ms/iteration', 'n_gpus': world, 'ranks_seen': ranks_seen, 'frames_total': n_local * world, 'dim': D,
                          'components': M, 'higher_is_better': False, 'scaling': 'strong', 'dtype': 'f64',
                          'kmeans_init_ms': km_s * 1e3, 'kmeans_lloyd_iterations': km_iters,
                          'k_fit_logprob_ms': lp_ms, 'k_fit_cov_ms': cv_ms,
                          'logprob_tflops': lp_tf, 'cov_tflops': cv_tf,
                          'cov_note': 'cov_tflops counts the DENSE product 2 n M D(D+1)/2; since round 5 the kernel skips frame groups '
                                      'and tiles whose responsibilities are below max(2^-200, 2^-70 nk) (kwy_gmm_em_cov_stats_dev), so the '
                                      'figure is a dense-equivalent rate and may exceed the matrix peak; k_fit_logprob skips nothing',
                          'roofline': {'bound': 'mfma', 'kernel': 'k_fit_logprob', 'achieved': lp_tf, 'peak': peak,
                                       'unit': 'TFLOP/s', 'frac': lp_tf / peak if lp_tf else None, 'traffic': None,
                                       'note': 'useful flops 2 n M D(D+1)/2 per kernel (the padded 16x16 blocks on the '
                                               'diagonal add 10 % issued work); peak = 1024 SIMDs x 32 flop/cycle x '
                                               '2.4 GHz, measured 75.4 with tools/mfma_f64_bench.hip; k_fit_cov: '
                                               'cov_tflops / peak'},
                          'process_group': (dist.get_backend() + f', world size {world}') if grouped else None,
                          'all_reduce_bytes_per_iteration': 8 * (M * (D + 1) + M * D * D + 1)}))
    if grouped:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
