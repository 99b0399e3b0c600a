"""Multi-process (world_size 2, gloo, CPU) test of the sharding used by bench.py:
disjoint round-robin shards that cover every utterance, and the SUM/MAX
reductions that turn per-rank counts into the whole-job throughput."""
import os
import socket

import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, q):
    import torch.distributed as dist
    from kwiiyatta_amd.parallel import gather_frame_counts, shard_indices
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    mine = shard_indices(n_items, rank, world)
    frames = 2001 * len(mine)
    total, slowest = gather_frame_counts(frames, 1.0 + rank)
    q.put((rank, mine, total, slowest))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    world, n_items = 2, 7
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shards = [r[1] for r in res]
    assert sorted(shards[0] + shards[1]) == list(range(n_items))
    assert not set(shards[0]) & set(shards[1])
    for _, _, total, slowest in res:
        assert total == 2001 * n_items
        assert slowest == 2.0


def test_shard_indices_properties():
    from kwiiyatta_amd.parallel import shard_indices
    for n in (0, 1, 8, 256, 257):
        for w in (1, 2, 4, 8):
            allv = sorted(sum((shard_indices(n, r, w) for r in range(w)), []))
            assert allv == list(range(n))
            sizes = [len(shard_indices(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1
