#!/usr/bin/env python
"""Benchmark of kwiiyatta's per-utterance conversion hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: the process starts N ranks itself, one per GPU (bench_launch.py), unless it already runs under
     python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one batch of synthetic 48 kHz utterance
pairs that are already resident in HBM.  Default workload = BASELINE config 3:
analyse source and target (CheapTrick + D4C over given f0 tracks), pad with
freshly drawn silent spectra (numpy's legacy generator continued on the GPU),
sp2mc, DTW-align source onto target (FastDTW, radius 32), convert with a
64-component GMM (delta + MLPG), mc2sp, WORLD synthesis.  Pairs are independent:
the default driver runs a rank's pairs in LOCKSTEP through the batched entries
of include/kwy.h (kwiiyatta_amd.pipeline.PairBatchPipeline: two waves of 16
pairs on four streams, the whole step one HIP graph); ranks shard pairs with NO
collective on the data path (weak scaling: pairs per GPU fixed); only the
timing is reduced (MAX).  One frame = 5 ms of SOURCE audio (240 samples).
Rank 0 prints one JSON line.

--driver streams is round 3's driver (one stream and one graph per pair; it
needs GPU_MAX_HW_QUEUES, see below), --driver serial runs one wave of the
lockstep driver on ONE stream (per-kernel durations as rocprofv3 sees them).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _argv_value(flag, default):
    return sys.argv[sys.argv.index(flag) + 1] if flag in sys.argv[:-1] else default


# The stream-per-pair drivers (--driver streams, the utterance workloads) only overlap their streams if these land on
# different hardware queues; the runtime's default of 4 serialises the rest (round 3: 4 queues 1.43 M frames/s, 64:
# 1.91 M).  The variable must be in the environment before the HIP runtime starts.  The default lockstep driver uses
# four streams and leaves the environment alone.
if _argv_value('--driver', 'batch') == 'streams' or (_argv_value('--workload', 'pair') == 'utterance' and
                                                   int(_argv_value('--utterances', '0')) == 0):
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '64')

FS = 48000
FRAME_PERIOD = 5.0
# counter summaries of the current kernels (tools/final_measure.sh + tools/collect_profiles.sh)
PMC_TRAFFIC, PMC_SQ = 'r5_pmc_traffic.json', 'r5_pmc_sq_summary.json'
METRIC = 'frames/sec end-to-end analyse->align->convert->synth, 48 kHz 5 ms hop'


def pair_silence(index, K=1025):
    """The four pad spectra of pair `index` (source head, source tail, target head, target tail): what pad_silence
    draws (kwiiyatta/vocoder/world.py:158-161), from a generator seeded per pair so that the pipeline and the CPU
    chain see the same blocks."""
    rng = np.random.RandomState(1000 + index)
    return [np.abs(rng.normal(0, 2.220446049250313e-16 / FS, (100, K))) for _ in range(4)]


def cpu_pair_once(src, tgt, gmm_params, silence):
    """One pass of the config-3 path for one pair on the CPU oracle (oracle/chain.py: C restatement of the reference's
    pyworld / pysptk / fastdtw / nnmnkwii path, every stage fed by the oracle's own previous output), one thread.
    Returns the chain's result dict ('frames', 'seconds', 'wave', 'path', ...)."""
    from oracle import chain
    return chain.pair_chain(src, tgt, gmm_params, FS, silence)


def cpu_worker(path):
    """`bench.py --cpu-worker FILE`: a fresh process (no torch, no GPU) that runs the pair stored in FILE once and
    prints its wall time; the all-core figure starts one of these per host core."""
    d = np.load(path)
    r = cpu_pair_once((d['xs'], d['f0s'], d['ts']), (d['xt'], d['f0t'], d['tt']),
                      (d['weights'], d['means'], d['covs']), list(d['silence']))
    print(json.dumps({'frames': r['frames'], 'seconds': r['seconds']}))


def cpu_baseline_pair(src, tgt, gmm, silence, budget_s=25.0):
    """The CPU oracle on the host cores of this box: the FULL 10 s + 11 s pair of the benchmark workload, once on
    one core, then once per core on all cores at the same time (utterance-parallel, one process each).
    Returns (cpu_baseline object, the chain's result of the one-core pass)."""
    import subprocess
    import tempfile
    params = (gmm.weights_, gmm.means_, gmm.covariances_)
    ref = cpu_pair_once(src, tgt, params, silence)
    frames, sec1 = ref['frames'], ref['seconds']
    out = {'value': frames / sec1, 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
           'sample': f'one full pair of the workload ({frames} source frames: 10 s source + 11 s target) through the same '
                     f'analyse->align->convert->synth path on oracle/liboracle.so (C restatement of pyworld 0.2.8 / '
                     f'pysptk / fastdtw / nnmnkwii), 1 thread, {sec1:.1f} s'}
    # a one-GPU box hands this job 16 of its host cores (os.cpu_count() reports the whole machine)
    nproc = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)
    if sec1 * 1.5 <= budget_s and nproc > 1:
        with tempfile.TemporaryDirectory() as tmp:
            f = os.path.join(tmp, 'pair.npz')
            np.savez(f, xs=src[0], f0s=src[1], ts=src[2], xt=tgt[0], f0t=tgt[1], tt=tgt[2], weights=params[0],
                     means=params[1], covs=params[2], silence=np.stack(silence))
            t0 = time.perf_counter()
            procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', f],
                                      stdout=subprocess.PIPE, env=dict(os.environ, OMP_NUM_THREADS='1'))
                     for _ in range(nproc)]
            done = [json.loads(p.communicate()[0].decode().strip().splitlines()[-1]) for p in procs]
            wall = time.perf_counter() - t0
        out['all_cores'] = {'value': sum(d['frames'] for d in done) / wall, 'unit': 'frames/s', 'cores': nproc,
                            'processes': nproc, 'wall_seconds': wall,
                            'sample': f'{nproc} processes (os.cpu_count() = {os.cpu_count()}, CPU share of a one-GPU job: 16), the same full pair each, started '
                                      f'together; wall time includes process start-up'}
    return out, ref


def _make_utterance_job(job):
    """(seed, seconds[, keyword arguments of make_utterance]) -> utterance; module-level so that a process pool can
    run it.  synthetic.py is loaded by path: the workers need neither the package nor the HIP runtime."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('kwy_synthetic', os.path.join(ROOT, 'kwiiyatta_amd', 'synthetic.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seed, seconds = job[0], job[1]
    kw = job[2] if len(job) > 2 else {'f0_base': 110.0 + (seed % 7) * 15.0}
    return mod.make_utterance(seed=seed, fs=FS, seconds=seconds, **kw)


def host_generate(jobs):
    """The synthetic signals of `jobs` (see _make_utterance_job), made by a pool of FRESH worker processes (spawn: no
    torch, no HIP) on the host cores this rank may use.  Call before the process touches the GPU where possible."""
    import concurrent.futures as cf
    import multiprocessing as mp
    if not jobs:
        return []
    nproc = max(1, min(len(os.sched_getaffinity(0)), 16, len(jobs)))
    if nproc == 1:
        return [_make_utterance_job(j) for j in jobs]
    with cf.ProcessPoolExecutor(nproc, mp_context=mp.get_context('spawn')) as ex:
        return list(ex.map(_make_utterance_job, jobs, chunksize=max(1, min(4, len(jobs) // nproc))))


def pair_jobs(index, seconds):
    """the two generator jobs of pair `index` (source: 10 s; target: the same schedule warped to 11 s with shifted formants)"""
    return [(1234 + 2 * index, seconds, {}), (4321 + 2 * index, seconds, {'time_warp': 1.1, 'formant_scale': 1.12})]


def init_group(args):
    """(rank, local_rank, world, dev, rdev): the rank's place as torch.distributed.run / bench_launch set it, the
    process group started (RCCL under `nccl`), the device chosen."""
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
    dev = torch.device('cuda', local_rank)
    rdev = dev if args.backend == 'nccl' else torch.device('cpu')   # where the timing reductions live
    return rank, local_rank, world, dev, rdev


def reduce_ranks(rank, world, rdev, seconds, frames):
    """MAX of the elapsed time, SUM of the frames, the number of ranks that answered (a SUM of ones over the group:
    `ranks_seen`) and every rank's own frames per second of its own elapsed time."""
    if world == 1:
        return seconds, float(frames), 1, [frames / seconds]
    import torch
    import torch.distributed as dist
    tt = torch.tensor([seconds], dtype=torch.float64, device=rdev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    v = torch.zeros(world + 2, dtype=torch.float64, device=rdev)
    v[0], v[1], v[2 + rank] = float(frames), 1.0, frames / seconds
    dist.all_reduce(v, op=dist.ReduceOp.SUM)
    v = v.cpu().tolist()
    return float(tt.item()), v[0], int(round(v[1])), v[2:]


def config4_seeds(total, rank, world):
    """BASELINE config 4 shards ONE batch of `total` utterances round-robin: rank r takes r, r + W, ... (the
    reference's per-file loop, /root/reference/kwiiyatta/convert_voice.py:17-32); seed = global index."""
    return list(range(rank, total, world))


def config4_measure(args, grp, utts, steps, warmup):
    """Time `steps` passes of this rank's share of the config-4 batch (analyse + resynthesise every utterance, in
    lockstep waves of 16 or through a stream pool) under the contract's protocol.  Returns a dict (all ranks)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world, dev, rdev = grp
    from kwiiyatta_amd import corpus as cp
    resident = [tuple(torch.from_numpy(a).to(dev) for a in u) for u in utts]
    lockstep = args.driver != 'streams'
    drv = dict(driver='lockstep', lockstep=cp._Lockstep(local_rank)) if lockstep else \
        dict(driver='streams', pool=cp.StreamPool(local_rank, args.batch))
    ylen = [int(cp.lib.kwy_synth_length(len(u[1]), FRAME_PERIOD, FS)) for u in utts]
    out = [torch.empty(n, dtype=torch.float64, device=dev) for n in ylen]

    def step():
        return cp.resynthesize_batch(resident, FS, device_index=local_rank, out=out, **drv)[1] if resident else 0

    frames_step = 0
    for _ in range(max(1, warmup)):
        frames_step = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    el, frames_total, ranks_seen, per_rank = reduce_ranks(rank, world, rdev, el, float(frames_step * steps))
    finite = all(bool(torch.isfinite(w).all().item()) for w in out)
    first = [w.clone() for w in out]
    step()
    torch.cuda.synchronize()
    identical = all(torch.equal(a, b) for a, b in zip(first, out))
    return {'seconds': el, 'frames_total': frames_total, 'ranks_seen': ranks_seen, 'per_rank_frames_per_s': per_rank,
            'finite': finite, 'identical': identical, 'lockstep': lockstep, 'first': first, 'out': out}


def main_batch(args):
    """BASELINE config 4: ONE batch of `--utterances` distinct synthetic 48 kHz utterances (256 = the configuration's
    batch; seeds = global utterance index) sharded round-robin over the ranks, analysed and resynthesised in lockstep
    waves of 16 (or through a fixed pool of `--batch` streams) -- inputs resident in HBM, one frame = 5 ms of audio.
    Strong scaling: the batch is fixed, a rank takes 1/N of it.
    Reference flow: kwiiyatta/resynthesize_voice.py:46-79 per file."""
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    seeds = config4_seeds(args.utterances, rank, world)
    # host-side signal generation first, in worker processes, before this process touches the GPU
    utts = host_generate([(sd, args.seconds) for sd in seeds])

    import torch
    import torch.distributed as dist
    grp = init_group(args)
    rank, local_rank, world, dev, rdev = grp
    from kwiiyatta_amd import pipeline as pl
    m = config4_measure(args, grp, utts, args.steps, args.warmup)
    el, frames_total, finite, identical, lockstep, first = (m['seconds'], m['frames_total'], m['finite'], m['identical'],
                                                             m['lockstep'], m['first'])
    if rank == 0:
        # the D4C stage (dominant whole-chip kernels) of one utterance alone, HIP events on its stream
        lone = pl.UtterancePipeline(local_rank, FS, utts[0])
        lone.run(); lone.sync()
        lone.profile(True)
        per = {}
        for _ in range(7):
            lone.run(); lone.sync()
            for nme in ('k_d4c_body', 'k_d4c_bands', 'k_cheaptrick', 'k_syn_pulse'):
                ms, n = lone.profile_read(nme)
                if n:
                    per.setdefault(nme, []).append(ms / n)
        lone.profile(False)
        alone = {k: sorted(v)[len(v) // 2] for k, v in per.items()}
        T, K = lone.T, lone.K
        hop = FS * FRAME_PERIOD / 1000.0
        d4c_ms = alone.get('k_d4c_body', float('nan')) + alone.get('k_d4c_bands', float('nan'))
        bytes_per_launch = T * (hop * 8 + 16 + K * 8)
        achieved = bytes_per_launch / (d4c_ms * 1e-3) / 1e9
        value = frames_total / el
        res = {
            'metric': METRIC, 'value': value, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1000.0 * el / args.steps, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'ranks_seen': m['ranks_seen'], 'per_rank_frames_per_s': m['per_rank_frames_per_s'],
            'config': {'workload': f'config4: ONE batch of {args.utterances} distinct synthetic 48 kHz {args.seconds:g} s '
                                   f'utterances sharded round-robin over the ranks (T={T}, K={K}; seeds = global utterance index): CheapTrick + D4C + '
                                   f'WORLD synthesis each' + (', in waves of 16 through the batched entries on two streams'
                                                              if lockstep else ', through a fixed pool of streams'),
                       'utterances_total': args.utterances, 'utterances_this_rank': len(utts),
                       'streams_per_gpu': 2 if lockstep else args.batch,
                       'GPU_MAX_HW_QUEUES': os.environ.get('GPU_MAX_HW_QUEUES'),
                       'launch': ('kwiiyatta_amd.corpus.ConvertWave: one launch per stage and wave (kwy_*_batch_dev); inputs '
                                  'resident in HBM; every waveform kept') if lockstep else
                                 ('per stream one pipeline per utterance shape; its pass is a captured HIP graph from the '
                                  'second utterance of that shape on; inputs resident in HBM, copied device-to-device '
                                  'into the pipeline\'s buffers; every waveform kept'),
                       'parallelism': (f'utterances in lockstep waves of 16, utterance-per-GPU x{world}, no collective' if lockstep
                                       else f'utterance-per-stream x{args.batch}, utterance-per-GPU x{world}, no collective')},
            'real_time_factor': value / 200.0,
            'hbm_fraction_whole_path': value / world * 36664 / 8e12,
            'kernel_ms_per_launch_alone': alone,
            'roofline': {'bound': 'hbm', 'kernel': 'k_d4c_body+k_d4c_bands', 'achieved': achieved, 'peak': 8000.0,
                         'unit': 'GB/s', 'frac': achieved / 8000.0, 'traffic': None, 'avg_launch_ms': d4c_ms,
                         'algorithmic_bytes_per_launch': bytes_per_launch,
                         'note': 'one utterance alone, HIP events on its stream, median of 7 passes'},
            'checks': {'all_outputs_finite': finite, 'second_pass_bit_identical': identical},
            'cpu_baseline': None,
        }
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle as ko
            x, f0, t = utts[0]
            t1 = time.perf_counter()
            sp = ko.cheaptrick(x, f0, t, FS)
            apv = ko.d4c(x, f0, t, FS)
            ref = ko.synthesize(f0, sp, apv, FS, FRAME_PERIOD)
            sec = time.perf_counter() - t1
            res['cpu_baseline'] = {'value': len(f0) / sec, 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
                                   'sample': f'utterance 0 of the batch ({len(f0)} frames): cheaptrick + d4c + synthesize on '
                                             f'oracle/liboracle.so, 1 thread, {sec:.1f} s'}
            res['parity'] = {'wave_rms_vs_cpu_chain': float(np.sqrt(np.mean((first[0].cpu().numpy() - ref) ** 2))),
                             'tolerance': 1e-4, 'note': 'utterance 0 against the all-oracle analyse -> synthesise chain'}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def main_lockstep(args):
    """BASELINE config 3 on the lockstep driver (kwiiyatta_amd.pipeline.PairBatchPipeline): a rank's pairs as waves of
    <= 16 through the batched entries of include/kwy.h, the step one HIP graph on four streams; the pad spectra of
    every pair are drawn INSIDE the step from numpy's legacy generator continued on the device (what `align` does per
    call: /root/reference/kwiiyatta/vocoder/feature.py:19-41, world.py:158-161)."""
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    serial = args.driver == 'serial'
    # ---- host-side signal generation first, in fresh worker processes, before this process touches the GPU
    mine = list(range(rank, world * args.batch, world))       # = parallel.shard_indices: pair i belongs to rank i % W
    nbase = min(len(mine), args.distinct)      # distinct host-generated signal pairs (default: every pair of the batch)
    c4_total = 0 if (serial or args.config4 == 'off') else args.config4_utterances
    c4_seeds = config4_seeds(c4_total, rank, world)
    made = host_generate(sum((pair_jobs(mine[i], args.seconds) for i in range(nbase)), []) +
                         [(sd, args.seconds) for sd in c4_seeds])
    base = [(made[2 * i], made[2 * i + 1]) for i in range(nbase)]
    c4_utts = made[2 * nbase:]

    import torch
    import torch.distributed as dist
    grp = init_group(args)
    rank, local_rank, world, dev, rdev = grp

    from kwiiyatta_amd import pipeline as pl
    from kwiiyatta_amd.backend.nprandom import DeviceRandomState

    gmm = pl.synthetic_gmm(order=24, components=args.components, seed=0)
    dgmm = pl.DeviceGMM(gmm.weights_, gmm.means_, gmm.covariances_, dev)
    pairs = [base[i % nbase] for i in range(len(mine))]
    seed = 1000 + rank
    rng = DeviceRandomState.from_seed(seed, device_index=local_rank)
    wav = args.workload == 'wav'            # wav in -> int16 out: DIO + StoneMask and the post-step inside the step
    pipe = pl.PairBatchPipeline(local_rank, FS, pairs, dgmm, waves=1 if serial else args.waves, rng=rng, serial=serial,
                                wav_in=wav, pcm=wav, chain_priority=args.chain_priority == 'on')
    torch.cuda.synchronize()

    def sync_all():
        pipe.sync()
        torch.cuda.synchronize()

    def reduce_max(seconds):
        if world > 1:
            tt = torch.tensor([seconds], dtype=torch.float64, device=rdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            seconds = float(tt.item())
        return seconds

    def timed(step_fn, steps, finish=None):
        """barrier + synchronize, `steps` steps, synchronize + barrier; MAX over ranks"""
        sync_all()
        if finish:
            finish()
        if world > 1:
            dist.barrier()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn()
        sync_all()
        if finish:
            finish()
        if world > 1:
            dist.barrier()
        return reduce_max(time.perf_counter() - t0)

    if args.graph:
        pipe.capture()
        step = pipe.replay
    else:
        step = pipe.run
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
    el_own = time.perf_counter() - t0
    el, frames_total, ranks_seen, per_rank = reduce_ranks(rank, world, rdev, el_own, float(pipe.frames * args.steps))
    value = frames_total / el

    # One more (untimed) pass with the generator's state read before it: numpy, set to that state, must draw the same
    # pads for all pairs and end in the same state.  Pair 0's pads, DTW path and waveform of that pass feed the parity
    # check against the CPU chain below.
    state_before = rng.get_state()
    step()
    sync_all()
    draws_equal = None
    if rank == 0:
        ref = np.random.RandomState()
        ref.set_state(state_before)
        worst = 0.0
        for blk in pipe.pad_rows:
            exp = np.abs(ref.normal(0, pl.EPS / FS, (pl.PAD_LEN, pipe.K)))
            worst = max(worst, float(np.abs(blk.cpu().numpy() - exp).max() / exp.max()))
        st_d, st_n = rng.get_state(), ref.get_state()
        draws_equal = {'state_equals_numpy': bool(np.array_equal(st_d[1], st_n[1]) and st_d[2:] == tuple(st_n[2:])),
                       'pads_max_rel_diff_vs_numpy': worst, 'blocks': len(pipe.pad_rows),
                       'note': 'one extra pass after the timed region: np.random.RandomState set to the device '
                               'generator\'s state before the pass draws the same ' + str(len(pipe.pad_rows)) +
                               ' blocks (values to the rounding of log) and ends in the same state'}
    pads0 = [blk.cpu().numpy() for blk in pipe.pad_rows[:4]]
    path_t, plen_t, _ = pipe.path(0)
    path0 = [tuple(r) for r in path_t.cpu().numpy()[:int(plen_t.item())].tolist()]
    wave0 = pipe.wave(0).cpu().numpy()

    # ---- per-kernel durations: the same step enqueued kernel by kernel right after the timed region (HIP events
    #      cannot be recorded inside a captured graph here), the library's events around its tracked kernels
    names = ['k_cheaptrick', 'k_d4c_lovetrain', 'k_d4c_body', 'k_d4c_bands', 'k_syn_phase', 'k_syn_pulse', 'k_syn_ola', 'k_sp2mc',
             'k_mc2sp', 'k_dtw_dist', 'k_dtw_values', 'k_dtw_codes', 'k_dtw_trace', 'k_dtw_small', 'k_gmm_logp', 'k_mlpg_chunks',
             'k_mlpg_finish', 'k_np_words', 'k_np_jump', 'k_np_words_seg', 'k_np_emit', 'k_cep2mc', 'k_dio_filter', 'k_dio_zc',
             'k_dio_candidates', 'k_dio_fix', 'k_stonemask', 'k_finish']
    pipe.profile(True)
    for _ in range(min(args.steps, 3)):
        pipe.run()
    sync_all()
    pipe.profile(False)
    kernel_ms = {}
    for nme in names:
        ms, n = pipe.profile_read(nme)
        if n:
            kernel_ms[nme] = [ms, n]

    # ---- variants (every rank takes part): pads replayed instead of drawn; transfers inside the step
    variants = {}
    if not args.no_variants and not serial:
        pipe.rng = None                       # the pads of the last pass stay in place
        if args.graph:
            pipe.capture()
        elp = timed(pipe.replay if args.graph else pipe.run, args.steps)
        variants['pads_replayed'] = {'ms_per_step': 1000.0 * elp / args.steps, 'frames_per_s_rank': pipe.frames * args.steps / elp,
                                     'note': 'the same step WITHOUT the draw (the pad rows keep the last draw): what the '
                                             'fresh pads cost is the difference to `value`'}
        pipe.rng = rng
        if args.graph:
            pipe.capture()
        feeder = pl.BatchHostFeeder(pipe)
        launch = (lambda p: p.replay()) if args.graph else (lambda p: p.run())
        feeder.step(launch)
        elp = timed(lambda: feeder.step(launch), args.steps, feeder.sync)
        variants['with_pcie'] = {'ms_per_step': 1000.0 * elp / args.steps, 'frames_per_s_rank': pipe.frames * args.steps / elp,
                                 'bytes_per_pair': int((feeder.host_in.numel() + feeder.host_out[0].numel()) * 8 // len(pairs)),
                                 'note': 'same steps with all waveforms uploaded from pinned host memory and the synthesised '
                                         'waveforms downloaded inside the timed region: one upload and one download per step on '
                                         'two service streams, two staging slots each way '
                                         '(kwiiyatta_amd.pipeline.BatchHostFeeder); never `value`'}
        del feeder
        if not wav:
            # the same step from WAVEFORMS to 16-bit samples: f0 extraction (DIO + StoneMask of both sides) at the head
            # of every wave, the post-step of synthesize/save and the int16 truncation at its end
            pw = pl.PairBatchPipeline(local_rank, FS, pairs, dgmm, waves=args.waves, wav_in=True, pcm=True,
                                      rng=DeviceRandomState.from_seed(seed + 500, device_index=local_rank))
            stepw = pw.run
            if args.graph:
                pw.run(); pw.sync()
                pw.capture()
                stepw = pw.replay
            for _ in range(3):
                stepw()
            pw.sync()
            torch.cuda.synchronize()
            elp = timed(stepw, args.steps, pw.sync)
            f0_dev = pw.waves[0].f0[0].cpu().numpy()
            f0_gen = np.asarray(pairs[0][0][1])
            both = (f0_dev > 0) & (f0_gen > 0)
            variants['wav_in_pcm_out'] = {
                'ms_per_step': 1000.0 * elp / args.steps, 'frames_per_s_rank': pw.frames * args.steps / elp,
                'ratio_to_value': (pw.frames * args.steps / elp) / (pipe.frames * args.steps / el_own),
                'dio_status_words_nonzero': int(np.count_nonzero(pw.f0_status())),
                'f0_vs_generating_contour': {'voicing_agreement': float(np.mean((f0_dev > 0) == (f0_gen > 0))),
                                             'median_rel_diff_voiced': float(np.median(np.abs(f0_dev[both] - f0_gen[both]) / f0_gen[both]))
                                             if both.any() else None},
                'pcm_peak': int(pw.pcm16(0).abs().max().item()),
                'note': 'the same pairs from WAVEFORMS to int16: kwy_dio_batch_dev + kwy_stonemask_batch_dev of both sides at '
                        'the head of every wave (the f0 tracks are no longer given), kwy_finish_pcm16_batch_dev at its end; '
                        '`bench.py --workload wav` runs this as the main measurement with per-kernel times'}
            del pw

    # ---- BASELINE config 4 beside config 3 (every rank takes part): ONE batch of 256 distinct utterances sharded
    #      round-robin over the ranks (strong scaling), analyse + resynthesise in lockstep waves of 16
    config4 = None
    if c4_total:
        m4 = config4_measure(args, grp, c4_utts, args.config4_steps, 2)
        v4 = m4['frames_total'] / m4['seconds']
        config4 = {'metric': METRIC, 'value': v4, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.config4_steps,
                   'ms_per_step': 1000.0 * m4['seconds'] / args.config4_steps, 'scaling': 'strong',
                   'ranks_seen': m4['ranks_seen'], 'per_rank_frames_per_s': m4['per_rank_frames_per_s'],
                   'real_time_factor': v4 / 200.0,
                   'config': {'workload': f'config4: ONE batch of {c4_total} distinct synthetic 48 kHz {args.seconds:g} s utterances '
                                          'sharded round-robin over the ranks (seed = global utterance index): CheapTrick + '
                                          'D4C + WORLD synthesis each, in lockstep waves of 16 (kwiiyatta_amd.corpus.ConvertWave), '
                                          'inputs resident in HBM, every waveform kept',
                              'utterances_total': c4_total, 'utterances_this_rank': len(c4_utts)},
                   'checks': {'all_outputs_finite': m4['finite'], 'second_pass_bit_identical': m4['identical']}}
        del m4

    if rank == 0:
        K = pipe.K
        hop = FS * FRAME_PERIOD / 1000.0
        # the kernels with the GPU to themselves: ONE wave (<= 16 pairs) on ONE stream, kernel after kernel
        lone_pairs = pairs[:min(16, len(pairs))]
        lone = pipe if serial else pl.PairBatchPipeline(local_rank, FS, lone_pairs, dgmm, waves=1, serial=True, wav_in=wav, pcm=wav,
                                                        rng=DeviceRandomState.from_seed(7, device_index=local_rank))
        lone.run(); lone.sync()
        lone.profile(True)
        per_pass, alone_n = {}, {}
        for _ in range(7):
            lone.run(); lone.sync()
            for nme in names:
                ms, n = lone.profile_read(nme)
                if n:
                    per_pass.setdefault(nme, []).append(ms / n)
                    alone_n[nme] = n
        lone.profile(False)
        alone_ms = {k: sorted(v)[len(v) // 2] for k, v in per_pass.items()}       # medians over the passes
        D4C = 'k_d4c_body+k_d4c_bands'
        if 'k_d4c_body' in alone_ms and 'k_d4c_bands' in alone_ms:
            alone_ms[D4C] = alone_ms['k_d4c_body'] + alone_ms['k_d4c_bands']
            alone_n[D4C] = alone_n['k_d4c_body']
        if 'k_d4c_body' in kernel_ms and 'k_d4c_bands' in kernel_ms:
            a, b = kernel_ms['k_d4c_body'], kernel_ms['k_d4c_bands']
            kernel_ms[D4C] = [a[0] + b[0], min(a[1], b[1])]
        lw = lone.waves[0]
        frames_both = float(sum(lw.T))                   # analysis: source + target frames of the lone wave
        frames_tgt = float(sum(lw.Tt))
        launches = {k: alone_n.get(k, 1) for k in (D4C, 'k_cheaptrick', 'k_d4c_lovetrain', 'k_syn_pulse')}
        fpl = {k: (frames_tgt if k == 'k_syn_pulse' else frames_both) / max(1, launches[k]) for k in launches}
        # ALGORITHMIC HBM bytes per launch of the whole-chip kernels: hop new samples + (f0, t) in,
        # one K-bin f64 row out per frame; the synthesis reads sp + ap rows and writes hop samples
        algo = {D4C: fpl[D4C] * (hop * 8 + 16 + K * 8), 'k_cheaptrick': fpl['k_cheaptrick'] * (hop * 8 + 16 + K * 8),
                'k_d4c_lovetrain': fpl['k_d4c_lovetrain'] * (hop * 8 + 16 + 8),
                'k_syn_pulse': fpl['k_syn_pulse'] * (2 * K * 8 + hop * 8)}
        cand = [k for k in algo if k in kernel_ms]
        dom = max(cand, key=lambda k: kernel_ms[k][0]) if cand else D4C
        traffic = None
        try:        # PMC-measured HBM bytes (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes) per launch
            with open(os.path.join(ROOT, 'profiles', PMC_TRAFFIC)) as fh:
                pmc = json.load(fh)
            traffic = sum(pmc['kernels'][k]['hbm_bytes_per_launch_raw'] for k in dom.split('+')) * fpl[dom] / pmc['frames_per_launch']
        except (OSError, KeyError, ValueError):
            pass
        achieved = algo[dom] / (alone_ms[dom] * 1e-3) / 1e9 if dom in alone_ms else None
        sh = kernel_ms.get(dom)
        roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s',
                    'frac': achieved / 8000.0 if achieved else None, 'traffic': traffic,
                    'avg_launch_ms': alone_ms.get(dom), 'launches': alone_n.get(dom, 0) * 7,
                    'algorithmic_bytes_per_launch': algo[dom], 'frames_per_launch': fpl[dom],
                    'shared': {'avg_launch_ms': sh[0] / sh[1] if sh else None, 'launches': sh[1] if sh else 0,
                               'measured': 'the step enqueued kernel by kernel on its four streams right after the timed '
                                           'region: the launch shares the chip with the other wave\'s kernels'},
                    'note': 'the launch analyses 16 utterances (8 pairs\' sources and targets) in one grid; avg_launch_ms: HIP '
                            'events on the launching stream, one wave of 16 pairs on ONE stream (median of 7 passes), which '
                            '`bench.py --driver serial` under rocprofv3 reproduces (profiles/); the stage is bound by f64 '
                            'FFT arithmetic in LDS, not by HBM -- the HBM fraction is reported as asked (DESIGN.md section 5)'}
        roofline_compute = None
        if D4C in alone_ms:
            voiced = float((lw.f0_all > 0).sum().item()) / max(1, alone_n[D4C])
            flop = voiced * 10 * 5 * 4096 * 12
            tf = flop / (alone_ms[D4C] * 1e-3) / 1e12
            roofline_compute = {'bound': 'f64 vector', 'kernel': D4C, 'achieved': tf, 'peak': 78.6, 'unit': 'TFLOP/s',
                                'frac': tf / 78.6, 'avg_launch_ms': alone_ms[D4C], 'frames_with_work_per_launch': voiced,
                                'algorithmic_flop_per_frame': 10 * 5 * 4096 * 12,
                                'note': 'FFT flop only (windows, noise, smoothing, selects not counted); frames with work = '
                                        'frames with f0 > 0 (upper bound of the frames that pass the gate)'}
            try:
                with open(os.path.join(ROOT, 'profiles', PMC_SQ)) as fh:
                    sq = json.load(fh)
                roofline_compute['valu_issue'] = {
                    k: {'busy_share_of_launch_per_simd': sq[k]['VALU_busy_per_SIMD'],
                        'valu_instructions_per_wavefront': sq[k]['VALU_insts_per_wave']}
                    for k in ('k_d4c_body', 'k_d4c_bands') if k in sq}
            except (OSError, KeyError, ValueError):
                pass
        out = {
            'metric': METRIC, 'value': value, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1000.0 * el / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {
                'workload': ('config3: 48 kHz source (10 s, T=2001) + target (11 s, T=2201) pairs: ' +
                             ('f0 by DIO+StoneMask of both (wav in), ' if wav else '') + 'CheapTrick+D4C of '
                             'both, pad with freshly drawn silent spectra, sp2mc, FastDTW(radius 32) align, GMM(%d comp, '
                             'D=144)+MLPG convert, mc2sp, WORLD synthesis' % args.components +
                             (', post-step + int16 (pcm out)' if wav else '')),
                'pairs_per_gpu': args.batch, 'source_frames_per_pair': pipe.waves[0].T[0],
                'driver': 'one stream, kernel after kernel (--driver serial)' if serial else
                          'lockstep: %d waves of <= 16 pairs, two streams per wave, batched entries (one grid per stage and '
                          'wave)' % len(pipe.waves),
                'streams_per_gpu': 1 if serial else 2 * len(pipe.waves),
                'GPU_MAX_HW_QUEUES': os.environ.get('GPU_MAX_HW_QUEUES'),
                'launch': 'the whole step is one captured HIP graph' if args.graph else 'one host launch per kernel',
                'pad_spectra': 'drawn INSIDE every step: 4 x 100 x 1025 values per pair from numpy\'s legacy generator '
                               '(MT19937 + polar Box-Muller) continued on the device, one call per step '
                               '(kwy_np_normal_blocks_dev); draws_check compares the generator with numpy afterwards',
                'gmm_model_prepared': 'once per converter (the per-mixture matrices depend on the GMM only)',
                'parallelism': f'pairs in lockstep x{args.batch} per GPU, pair-per-GPU x{world}, no collective'},
            'real_time_factor': value / 200.0,
            'hbm_fraction_whole_path': value / world * 81000 / 8e12,
            'kernel_ms_per_launch': {k: v[0] / v[1] for k, v in sorted(kernel_ms.items())},
            'kernel_ms_per_launch_alone': {k: v for k, v in sorted(alone_ms.items())},
            'roofline': roofline,
            'roofline_compute': roofline_compute,
            'draws_check': draws_equal,
            'pads_replayed': variants.get('pads_replayed'),
            'with_pcie': variants.get('with_pcie'),
            'wav_in_pcm_out': variants.get('wav_in_pcm_out'),
            'parity': None,
            'distinct_pairs_per_gpu': nbase,
            'ranks_seen': ranks_seen,
            'per_rank_frames_per_s': per_rank,
            'config4': config4,
            'cpu_baseline': None,
        }
        if not args.no_cpu_baseline and world == 1:
            # the CPU chain on the very pads the device drew for pair 0 in the last timed pass
            src0, tgt0 = base[0]
            if wav:     # the CPU chain starts from the waveforms too: the oracle's own DIO + StoneMask tracks
                from oracle import oracle as ko
                src0, tgt0 = ((u[0], ko.stonemask(u[0], *ko.dio(u[0], FS), FS), u[2]) for u in (src0, tgt0))
            out['cpu_baseline'], ref = cpu_baseline_pair(src0, tgt0, gmm, pads0)
            out['parity'] = {
                'wave_rms_vs_cpu_chain': float(np.sqrt(np.mean((wave0 - ref['wave']) ** 2))),
                'wave_peak': float(np.abs(ref['wave']).max()),
                'dtw_path_equal': path0 == ref['path'], 'dtw_path_cells': len(ref['path']),
                'tolerance': 1e-4,
                'note': 'pair 0 of the pass right after the timed region vs oracle/chain.py on the same waveforms, f0 tracks, GMM and the pad '
                        'blocks the device drew for that pass, every oracle stage fed by the oracle\'s own previous output'}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main():
    if len(sys.argv) == 3 and sys.argv[1] == '--cpu-worker':
        return cpu_worker(sys.argv[2])
    ap_ = argparse.ArgumentParser()
    ap_.add_argument('--gpus', type=int, default=1)
    ap_.add_argument('--steps', type=int, default=20)
    ap_.add_argument('--warmup', type=int, default=25,
                     help='untimed steps first (a box that has just started needs more than a few: its first ~0.3 s of '
                          'work run 5 %% slower -- clocks, first touches -- whatever the code does)')
    ap_.add_argument('--batch', type=int, default=32, help='utterance pairs per GPU per step')
    ap_.add_argument('--seconds', type=float, default=10.0, help='source utterance length')
    ap_.add_argument('--workload', choices=['pair', 'utterance', 'wav'], default='pair',
                     help='pair: BASELINE config 3 with the f0 tracks given (the headline); wav: the same from waveforms to '
                          '16-bit samples (DIO + StoneMask and the post-step inside the step); utterance: configs 2 / 4')
    ap_.add_argument('--utterances', type=int, default=0,
                     help='with --workload utterance: BASELINE config 4 -- this many DISTINCT utterances per GPU and step '
                          '(seeds = global utterance index, 256 = the configuration\'s batch on one GPU) through a fixed '
                          'pool of --batch streams (kwiiyatta_amd.corpus.resynthesize_batch); 0 = config 2 (one '
                          'pipeline per stream)')
    ap_.add_argument('--components', type=int, default=64)
    ap_.add_argument('--driver', choices=['batch', 'streams', 'serial'], default='batch',
                     help='pair workload: batch = lockstep through the batched entries (default); streams = one stream and '
                          'one graph per pair (round 3); serial = one wave of the lockstep driver on one stream')
    ap_.add_argument('--waves', type=int, default=2, help='lockstep driver: waves of <= 16 pairs side by side')
    ap_.add_argument('--chain-priority', choices=('on', 'off'), default='off',
                     help='lockstep driver: the streams of the serial chain (alignment, conversion, rendering) get a '
                          'higher stream priority than the aperiodicity streams')
    ap_.add_argument('--no-variants', action='store_true', help='lockstep driver: skip the pads-replayed and PCIe variants')
    ap_.add_argument('--no-cpu-baseline', action='store_true')
    ap_.add_argument('--distinct', type=int, default=32,
                     help='distinct synthetic signal pairs per rank (cycled over the batch when fewer than --batch; made '
                          'by a pool of host processes before the GPU is touched)')
    ap_.add_argument('--config4', choices=['on', 'off'], default='on',
                     help='pair workload, lockstep driver: after config 3 also time BASELINE config 4 (ONE batch of '
                          '--config4-utterances distinct utterances sharded round-robin over the ranks) and report it '
                          'as the `config4` object of the line')
    ap_.add_argument('--config4-utterances', type=int, default=256)
    ap_.add_argument('--config4-steps', type=int, default=3)
    ap_.add_argument('--no-pcie-variant', action='store_true',
                     help='skip the second timed loop that uploads the waveforms and downloads the result inside the step')
    ap_.add_argument('--side-stream', choices=['auto', 'on', 'off'], default='auto',
                     help='pair workload: D4C on a second stream beside the alignment (a latency option; auto = only '
                          'when one pair runs alone, i.e. --batch 1)')
    ap_.add_argument('--no-graph', dest='graph', action='store_false',
                     help='enqueue every kernel of a pass from the host (about 170 launches per pair) instead of '
                          'replaying the pass as a captured HIP graph; the per-kernel HIP events are then recorded '
                          'inside the timed region itself')
    ap_.add_argument('--gmm-prepare-per-pair', action='store_true',
                     help='redo the GMM-only precomputation of MLPG for every pair (the reference builds an MLPG '
                          'object per convert() call) instead of once per converter')
    ap_.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL; gloo lets '
                                                        'several ranks share one GPU when rehearsing the launch)')
    args = ap_.parse_args()
    import bench_launch
    bench_launch.maybe_launch(args.gpus)     # --gpus N > 1 outside a rank: become the launcher of N ranks (never returns)

    args.side_stream = args.side_stream == 'on' or (args.side_stream == 'auto' and args.batch == 1)
    if args.workload == 'utterance' and args.utterances > 0:
        return main_batch(args)
    if args.workload in ('pair', 'wav') and args.driver != 'streams':
        if args.driver == 'serial' and args.batch > 16:
            args.batch = 16
        return main_lockstep(args)

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    # global utterance indices of this rank (weak scaling: `batch` per rank); signals made before the GPU is touched
    mine = list(range(rank, world * args.batch, world))
    nbase = min(len(mine), args.distinct)      # distinct host-generated signal pairs, reused round-robin
    made = host_generate(sum((pair_jobs(mine[i], args.seconds) for i in range(nbase)), []))
    base = [(made[2 * i], made[2 * i + 1]) for i in range(nbase)]

    import torch
    import torch.distributed as dist
    rank, local_rank, world, dev, rdev = init_group(args)

    from kwiiyatta_amd import pipeline as pl

    gmm = pl.synthetic_gmm(order=24, components=args.components, seed=0) if args.workload == 'pair' else None
    dgmm = pl.DeviceGMM(gmm.weights_, gmm.means_, gmm.covariances_, dev) if gmm is not None else None

    pipes = []
    for i in range(len(mine)):
        src, tgt = base[i % nbase]
        if args.workload == 'pair':
            pipes.append(pl.PairPipeline(local_rank, FS, src, tgt, dgmm, prepare_gmm_per_run=args.gmm_prepare_per_pair,
                                         side_stream=args.side_stream, silence=pair_silence(mine[i % nbase])))
        else:
            pipes.append(pl.UtterancePipeline(local_rank, FS, src))
    torch.cuda.synchronize()

    def step():
        for p in pipes:
            if args.graph:
                p.replay()
            else:
                p.run()

    def sync_all():
        for p in pipes:
            p.sync()
        torch.cuda.synchronize()

    if args.graph:
        try:
            for p in pipes:
                p.capture()
        except Exception as exc:     # a runtime that cannot capture: enqueue kernel by kernel instead
            sys.stderr.write(f'bench.py: HIP graph capture failed ({exc!r}); continuing with --no-graph\n')
            torch.cuda.synchronize()
            args.graph = False
    for _ in range(args.warmup):
        step()
    sync_all()
    if not args.graph:
        for p in pipes:
            p.profile(True)
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
        tt = torch.tensor([el], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())

    if args.graph:
        # Per-kernel durations for the roofline object: HIP events cannot be recorded inside a captured graph here
        # (hipEventRecord during capture: invalid resource handle), so the same passes are enqueued kernel by
        # kernel on the same streams right after the timed region, with the library's events around the tracked
        # kernels -- same kernels, same concurrency, not part of `value`.
        for p in pipes:
            p.profile(True)
        for _ in range(min(args.steps, 3)):
            for p in pipes:
                p.run()
        sync_all()
        for p in pipes:
            p.profile(False)

    def timed_variant(step_fn, finish=None, steps=None):
        """the timing protocol of the main loop (one untimed step, barrier, K steps, barrier, MAX over ranks) around
        another step function; returns seconds"""
        steps = args.steps if steps is None else steps
        step_fn()
        sync_all()
        if finish:
            finish()
        if world > 1:
            dist.barrier()
        tp = time.perf_counter()
        for _ in range(steps):
            step_fn()
        sync_all()
        if finish:
            finish()
        if world > 1:
            dist.barrier()
        elp = time.perf_counter() - tp
        if world > 1:
            tt = torch.tensor([elp], dtype=torch.float64, device=rdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elp = float(tt.item())
        return elp

    def launch(p):
        if args.graph:
            p.replay()
        else:
            p.run()

    # The variants below add service streams (copies, the pad generator) to the pair streams.  The chip runs 32 hardware
    # queues; with more live streams than that, streams share queues, and a copy that sits behind another pair's pass
    # in a shared queue while a third pair waits for it serialises the step (measured: HostFeeder with two extra
    # streams beside 32 pair streams 180 - 410 ms per step instead of 35).  So the variants run 30 pairs and use the
    # streams of the two idle pairs as service streams, and say so.
    vpipes = pipes[:30] if len(pipes) > 30 else pipes
    spare = [p.stream for p in pipes[30:32]] if len(pipes) >= 32 else [None, None]
    vframes = sum(p.frames for p in vpipes)

    def sync_v():
        for p in vpipes:
            p.sync()
        torch.cuda.synchronize()

    def run_pcie_variant():
        if args.workload != 'pair' or args.no_pcie_variant:
            return None
        feeder = pl.HostFeeder(vpipes, up=spare[0], down=spare[1])
        elp = timed_variant(lambda: feeder.step(launch), feeder.sync)
        pcie = {'ms_per_step': 1000.0 * elp / args.steps, 'pairs_per_step': len(vpipes),
                'frames_per_s_rank': vframes * args.steps / elp,
                'bytes_per_pair': int((feeder.host_in.numel() + feeder.host_out[0].numel()) * 8 // len(vpipes)),
                'note': 'same steps with the two waveforms uploaded from pinned host memory and the synthesised waveform '
                        'downloaded inside the timed region: one upload and one download per step (all pairs in one '
                        'block each way) on two service streams, two staging slots, so the transfers of neighbouring '
                        'steps overlap with the kernels (kwiiyatta_amd.pipeline.HostFeeder); never `value`'}
        del feeder
        return pcie

    def run_pad_variant():
        """`value` replays pad spectra that were drawn ONCE per pipeline, outside the timed region.  Here every pass
        gets fresh ones, as kwiiyatta.pad_silence draws them per call: (device) from numpy's legacy generator
        reproduced on the GPU, one step ahead of the pipelines (kwiiyatta_amd.pipeline.SilenceFeeder); (host) with
        np.random.normal itself, serial on one host thread, uploaded -- a few steps only, it is slow."""
        if args.workload != 'pair' or args.no_pcie_variant:
            return None
        from kwiiyatta_amd.backend.nprandom import DeviceRandomState
        frames_step = vframes
        feeder = pl.SilenceFeeder(vpipes, DeviceRandomState.from_seed(1000 + rank, device_index=local_rank,
                                                                      stream=spare[0]))
        eld = timed_variant(lambda: feeder.step(launch), feeder.sync)
        res = {'device': {'ms_per_step': 1000.0 * eld / args.steps, 'pairs_per_step': len(vpipes),
                          'frames_per_s_rank': frames_step * args.steps / eld,
                          'note': 'same steps with the four pad blocks of every pair drawn inside the timed region by '
                                  'k_np_normal (numpy\'s MT19937 + polar Box-Muller stream, draw for draw) on its own '
                                  'stream, double-buffered; never `value`'}}
        K = pipes[0].K
        stage = [[torch.empty((100, K), dtype=torch.float64).pin_memory() for _ in range(4)] for _ in vpipes]
        host_s = [0.0]
        host_steps = 2

        def step_pad():
            for p, st in zip(vpipes, stage):
                p.sync()                 # the pinned blocks of this pair are about to be overwritten
                t_ = time.perf_counter()
                drawn = [pl.draw_silence(FS, K) for _ in st]
                host_s[0] += time.perf_counter() - t_
                for blk, d_ in zip(st, drawn):
                    blk.copy_(torch.from_numpy(d_))        # (pinned host memory is slow to WRITE from the CPU here)
                rows = p.src.silence_rows() + p.tgt.silence_rows()
                with torch.cuda.stream(p.stream):
                    for dst, blk in zip(rows, st):
                        dst.copy_(blk, non_blocking=True)
                launch(p)
        elp = timed_variant(step_pad, steps=host_steps)
        res['host'] = {'ms_per_step': 1000.0 * elp / host_steps, 'steps': host_steps, 'pairs_per_step': len(vpipes),
                       'frames_per_s_rank': frames_step * host_steps / elp,
                       'host_draw_ms_per_pair': 1000.0 * host_s[0] / ((host_steps + 1) * len(vpipes)),
                       'note': 'same steps with the four pad blocks of every pair drawn on the host inside the timed '
                               'region (np.random.normal from the global legacy generator, one thread) and uploaded; '
                               'never `value`'}
        return res

    frames_rank = sum(p.frames for p in pipes) * args.steps
    if world > 1:
        ft = torch.tensor([frames_rank], dtype=torch.float64, device=rdev)
        dist.all_reduce(ft, op=dist.ReduceOp.SUM)
        frames_total = float(ft.item())
    else:
        frames_total = float(frames_rank)
    value = frames_total / el

    if rank == 0:
        T = pipes[0].frames
        K = pipes[0].K
        names = ['k_cheaptrick', 'k_d4c_lovetrain', 'k_d4c_body', 'k_d4c_bands', 'k_syn_phase', 'k_syn_pulse', 'k_syn_ola', 'k_sp2mc',
                 'k_mc2sp', 'k_dtw_dist', 'k_dtw_values', 'k_dtw_codes', 'k_dtw_trace', 'k_dtw_small', 'k_gmm_prep', 'k_gmm_logp', 'k_mlpg_chunks', 'k_mlpg_finish', 'k_align_project']
        kernel_ms = {}
        for p in pipes:
            for nme in names:
                ms, n = p.profile_read(nme)
                if n:
                    a = kernel_ms.setdefault(nme, [0.0, 0])
                    a[0] += ms
                    a[1] += n
        # the same kernels with the GPU to themselves: one pair, one stream, after the timed region
        # (per-launch durations inside the timed region include the time a kernel shares the CUs with
        # the other streams' kernels)
        alone_ms = {}
        if args.workload == 'pair':      # one kernel at a time: everything on one stream (no side stream for D4C)
            lone = pl.PairPipeline(local_rank, FS, *base[0], dgmm, prepare_gmm_per_run=args.gmm_prepare_per_pair,
                                   side_stream=False, silence=pair_silence(mine[0]))
        else:
            lone = pipes[0]
        lone.profile(True)
        lone.run(); lone.sync()
        for nme in names:
            lone.profile_read(nme)
        # median over passes: now and then a pass is held up for ~10 ms by the queue scheduler (seen in rocprofv3 traces
        # as one dispatch of an otherwise 0.04 ms kernel lasting 11 ms); a mean would carry that into the roofline
        alone_n = {}
        per_pass = {}
        for _ in range(7):
            lone.run(); lone.sync()
            for nme in names:
                ms, n = lone.profile_read(nme)
                if n:
                    per_pass.setdefault(nme, []).append(ms / n)
                    alone_n[nme] = alone_n.get(nme, 0) + n
        for nme, vals in per_pass.items():
            alone_ms[nme] = sorted(vals)[len(vals) // 2]
        # D4C's per-frame work is two launches (k_d4c_body: centroids, power spectrum, group delay -> one H+1 row of
        # scratch per frame; k_d4c_bands: band FFTs, selection, interpolation -> the K-bin row).  The roofline prices
        # the stage: both durations summed against the stage's algorithmic bytes and flop.
        D4C = 'k_d4c_body+k_d4c_bands'
        for tbl in (kernel_ms, alone_ms):
            if 'k_d4c_body' in tbl and 'k_d4c_bands' in tbl:
                a, b = tbl['k_d4c_body'], tbl['k_d4c_bands']
                tbl[D4C] = [a[0] + b[0], min(a[1], b[1])] if isinstance(a, list) else a + b
        if D4C in alone_ms:
            alone_n[D4C] = alone_n['k_d4c_body']
        # Frames one launch of a per-frame kernel processes.  Pair workload: the analysis kernels take BOTH utterances
        # of the pair in one grid (kwy_*_batch_dev: source + target frames per launch) unless D4C runs on the side
        # stream (--batch 1 in the timed region: one utterance per launch); the lone pipeline measured for the
        # roofline always batches.  The synthesis renders the target's frames.
        if args.workload == 'utterance':
            fpl = fpl_syn = float(T)
        else:
            fpl = float(pipes[0].src.T + pipes[0].tgt.T)
            fpl_syn = float(pipes[0].tgt.T)
        hop = FS * FRAME_PERIOD / 1000.0
        # ALGORITHMIC HBM bytes per launch of the whole-chip kernels:
        # hop new samples + (f0, t) in, one K-bin f64 row out per frame; synthesis reads sp+ap rows, writes hop samples.
        algo = {D4C: fpl * (hop * 8 + 16 + K * 8), 'k_cheaptrick': fpl * (hop * 8 + 16 + K * 8),
                'k_d4c_lovetrain': fpl * (hop * 8 + 16 + 8), 'k_syn_pulse': fpl_syn * (2 * K * 8 + hop * 8)}
        # The dominant kernel = the whole-chip kernel with the largest summed duration.  The single-workgroup
        # serial kernels (k_dtw_values, k_syn_phase, ...) occupy one CU each and overlap with other streams;
        # they bound latency, not throughput, and are listed in kernel_ms_per_launch.
        cand = [k for k in algo if k in kernel_ms]
        dom = max(cand, key=lambda k: kernel_ms[k][0]) if cand else D4C
        tot_ms, launches = kernel_ms.get(dom, (0.0, 0))
        bytes_per_launch = algo[dom]
        avg_s = (tot_ms / launches) * 1e-3 if launches else float('nan')
        # (with D4C on the side stream the timed passes launch the analysis per utterance: half the bytes per launch)
        shared_bytes = bytes_per_launch / 2 if (args.workload == 'pair' and args.side_stream and dom != 'k_syn_pulse') \
            else bytes_per_launch
        achieved = shared_bytes / avg_s / 1e9 if launches else None
        traffic = None
        try:        # PMC-measured HBM bytes (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes) per launch
            with open(os.path.join(ROOT, 'profiles', PMC_TRAFFIC)) as fh:
                pmc = json.load(fh)
            traffic = sum(pmc['kernels'][k]['hbm_bytes_per_launch_raw'] for k in dom.split('+')) \
                * fpl / pmc['frames_per_launch']
        except (OSError, KeyError, ValueError):
            pass
        # The roofline line is priced on the kernel's own duration: HIP events around launches of one stream with
        # the GPU to itself (they agree with rocprofv3's per-dispatch durations, profiles/r3_pair_b1_kernel_stats.csv).
        # Events around a launch that competes with 31 other streams also span the time the dispatch waits in its
        # hardware queue -- about 3x what rocprofv3 reports for the same dispatches -- and are kept in `shared`.
        alone_achieved = (bytes_per_launch / (alone_ms[dom] * 1e-3) / 1e9) if dom in alone_ms else None
        roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': alone_achieved, 'peak': 8000.0, 'unit': 'GB/s',
                    'frac': (alone_achieved / 8000.0) if alone_achieved else None, 'traffic': traffic,
                    'avg_launch_ms': alone_ms.get(dom), 'launches': alone_n.get(dom, 0),
                    'algorithmic_bytes_per_launch': bytes_per_launch, 'frames_per_launch': fpl if dom != 'k_syn_pulse' else fpl_syn,
                    'shared': {'avg_launch_ms': tot_ms / launches if launches else None, 'launches': launches,
                               'achieved': achieved,
                               'measured': ('passes enqueued kernel by kernel on the same streams right after the '
                                            'timed region (HIP events cannot be recorded inside the captured graphs '
                                            'the timed region replays)') if args.graph else 'inside the timed region'},
                    'note': 'kernel is bound by f64 FFT arithmetic and barrier latency in LDS, not by HBM; the HBM '
                            'fraction is reported as asked (DESIGN.md section 5).  traffic = FETCH_SIZE+WRITE_SIZE of '
                            'profiles/r3_pmc_traffic.json scaled to the frames of one launch.  avg_launch_ms: HIP events '
                            'on the launching stream, one stream running, right after the timed region'}
        # Compute roofline of the same kernel family: algorithmic f64 flop (5 N log2 N per real FFT of
        # size N; the D4C stage runs 10 transforms of 4096 per frame that passes the voicing gate) / kernel time /
        # the f64 vector peak (1024 SIMDs x 16 lanes x 2 flop x 2.4 GHz = 78.6 TFLOP/s).
        roofline_compute = None
        if args.workload == 'pair' and D4C in alone_ms:
            voiced = float((pipes[0].src.f0 > 0).sum().item()) + float((pipes[0].tgt.f0 > 0).sum().item())
            flop = voiced * 10 * 5 * 4096 * 12
            tf = flop / (alone_ms[D4C] * 1e-3) / 1e12
            roofline_compute = {'bound': 'f64 vector', 'kernel': D4C, 'achieved': tf, 'peak': 78.6,
                                'unit': 'TFLOP/s', 'frac': tf / 78.6, 'avg_launch_ms': alone_ms[D4C],
                                'frames_with_work_per_launch': voiced,
                                'algorithmic_flop_per_frame': 10 * 5 * 4096 * 12,
                                'note': 'FFT flop only (windows, RNG, smoothing, selects not counted); frames with '
                                        'work = frames with f0 > 0 (upper bound of the frames that pass the gate)'}
            try:    # what the stage is bound by: vector-instruction issue (SQ counters, tools/pmc_sq.sh, one utterance alone)
                with open(os.path.join(ROOT, 'profiles', PMC_SQ)) as fh:
                    sq = json.load(fh)
                roofline_compute['valu_issue'] = {
                    k: {'busy_share_of_launch_per_simd': sq[k]['VALU_busy_per_SIMD'],
                        'valu_instructions_per_wavefront': sq[k]['VALU_insts_per_wave']}
                    for k in ('k_d4c_body', 'k_d4c_bands') if k in sq}
            except (OSError, KeyError, ValueError):
                pass
        # the kernel with the largest SUMMED duration of all (what a rocprofv3 --stats table puts first)
        single = {k: v for k, v in kernel_ms.items() if '+' not in k}
        top = max(single, key=lambda k: single[k][0]) if single else None
        by_sum = None
        if top is not None:
            by_sum = {'kernel': top, 'avg_launch_ms': kernel_ms[top][0] / kernel_ms[top][1],
                      'launches': kernel_ms[top][1], 'alone_avg_launch_ms': alone_ms.get(top),
                      'share_of_tracked_kernel_time': kernel_ms[top][0] / sum(v[0] for v in single.values()),
                      'note': ('one workgroup per launch: a serial recurrence (FastDTW recurrence / back-trace, critical path '
                               'Tx+Ty steps) that holds 1 of 256 CUs and overlaps with the other streams; it bounds the '
                               'latency of one pair, not the throughput, and has no HBM or MFMA roofline')
                              if top in ('k_dtw_values', 'k_dtw_trace', 'k_syn_phase', 'k_align_project') else
                              'whole-chip kernel'}
        # whole-path algorithmic bytes per source frame
        path_bytes = 36664 if args.workload == 'utterance' else 81000
        out = {
            'metric': METRIC, 'value': value, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1000.0 * el / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {
                'workload': ('config3: 48 kHz source (10 s, T=2001) + target (11 s, T=2201) pairs: CheapTrick+D4C of '
                             'both, pad, sp2mc, FastDTW(radius 32) align, GMM(%d comp, D=144)+MLPG convert, mc2sp, '
                             'WORLD synthesis' % args.components) if args.workload == 'pair' else
                            'config2: 48 kHz 10 s utterances (T=2001, K=1025): CheapTrick + D4C + WORLD synthesis',
                'pairs_per_gpu' if args.workload == 'pair' else 'utterances_per_gpu': args.batch,
                'source_frames_per_pair': T, 'streams_per_gpu': args.batch,
                'd4c_on_side_stream': bool(args.side_stream) if args.workload == 'pair' else None,
                'gmm_model_prepared': ('per pair' if args.gmm_prepare_per_pair else
                                       'once per converter (the per-mixture matrices depend on the GMM only)')
                if args.workload == 'pair' else None,
                'launch': ('one captured HIP graph per pass and stream; per-kernel HIP-event durations from the same '
                           'passes enqueued kernel by kernel right after the timed region') if args.graph else
                          'one host launch per kernel; per-kernel HIP events inside the timed region',
                'pad_spectra': 'drawn once per pipeline, outside the timed region; with_pad_draw times the same steps '
                               'with fresh pads per pass (device generator / host numpy)'
                if args.workload == 'pair' else None,
                'parallelism': f'utterance-per-stream x{args.batch}, utterance-per-GPU x{world}, no collective'},
            'real_time_factor': value / 200.0,
            'hbm_fraction_whole_path': value / world * path_bytes / 8e12,
            'kernel_ms_per_launch': {k: v[0] / v[1] for k, v in sorted(kernel_ms.items())},
            'kernel_ms_per_launch_alone': {k: v for k, v in sorted(alone_ms.items())},
            'roofline': roofline,
            'roofline_compute': roofline_compute,
            'with_pcie': None,
            'with_pad_draw': None,
            'parity': None,
            'distinct_pairs_per_gpu': nbase,
            'largest_summed_kernel': by_sum,
            'cpu_baseline': None,
        }
    # the variants come last: every rank takes part, and the per-kernel measurements above are done
    pcie = run_pcie_variant()
    pad = run_pad_variant()
    if pad:                               # pass again with the original pads: pipes[0].wave is what the parity check reads
        for p_, i_ in zip(pipes, range(len(pipes))):
            rows = p_.src.silence_rows() + p_.tgt.silence_rows()
            with torch.cuda.stream(p_.stream):
                for dst, sil in zip(rows, pair_silence(mine[i_ % nbase])):
                    dst.copy_(torch.from_numpy(np.ascontiguousarray(sil)))
        step()
        sync_all()
    if rank == 0:
        if pcie:
            out['with_pcie'] = pcie
        if pad:
            out['with_pad_draw'] = pad
        if not args.no_cpu_baseline and world == 1:
            if args.workload == 'pair':
                out['cpu_baseline'], ref = cpu_baseline_pair(base[0][0], base[0][1], gmm, pair_silence(mine[0]))
                # the same pair, the same pads: the pipeline's final waveform against the all-CPU chain
                p0 = pipes[0]
                n = int(p0.path_len.item())
                path = [tuple(r) for r in p0.path.cpu().numpy()[:n].tolist()]
                wave = p0.wave.cpu().numpy()
                out['parity'] = {
                    'wave_rms_vs_cpu_chain': float(np.sqrt(np.mean((wave - ref['wave']) ** 2))),
                    'wave_peak': float(np.abs(ref['wave']).max()),
                    'dtw_path_equal': path == ref['path'], 'dtw_path_cells': len(ref['path']),
                    'tolerance': 1e-4,
                    'note': 'pair 0 of the timed batch after the last pass vs oracle/chain.py on the same waveforms, f0 '
                            'tracks, GMM and pad blocks, every oracle stage fed by the oracle\'s own previous output'}
            else:
                from oracle import oracle as ko
                x, f0, t = base[0][0]
                n, Tc = int(FS * 2.0), 401
                xs, f0s, ts = np.ascontiguousarray(x[:n]), np.ascontiguousarray(f0[:Tc]), np.ascontiguousarray(t[:Tc])
                reps, t1 = 0, time.perf_counter()
                while time.perf_counter() - t1 < 15 and reps < 8:
                    sp = ko.cheaptrick(xs, f0s, ts, FS)
                    apv = ko.d4c(xs, f0s, ts, FS)
                    ko.synthesize(f0s, sp, apv, FS, FRAME_PERIOD)
                    reps += 1
                out['cpu_baseline'] = {'value': reps * Tc / (time.perf_counter() - t1), 'unit': 'frames/s',
                                       'cores': 1, 'kind': 'port',
                                       'sample': f'{reps}x first 2 s (401 frames): cheaptrick+d4c+synthesize, '
                                                 f'oracle/liboracle.so, 1 thread'}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
