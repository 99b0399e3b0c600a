"""Multi-GPU sharding of the conversion path: one process per GPU, utterances
(or source/target pairs) are independent units, so ranks simply take disjoint
index sets -- there is NO collective on the data path.
torch.distributed (RCCL on ROCm, gloo on CPU) is used only to agree on counts
and timings."""


def shard_indices(n_items, rank, world_size):
    """Round-robin assignment: item i belongs to rank i % world_size."""
    if not (0 <= rank < world_size):
        raise ValueError(f'rank {rank} outside world of size {world_size}')
    return list(range(rank, n_items, world_size))


def gather_frame_counts(local_frames, local_seconds):
    """(total frames over all ranks, slowest rank's time) via all_reduce SUM / MAX.
    Works with any initialised backend; without a process group it is the identity."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(local_frames), float(local_seconds)
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    f = torch.tensor([float(local_frames)], dtype=torch.float64, device=dev)
    t = torch.tensor([float(local_seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(f.item()), float(t.item())
