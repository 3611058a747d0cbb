"""Sharding of independent renders over ranks (one process per GPU).

The path has no exchange step: every render is independent, so ranks take
contiguous blocks of renders and only the {frames, checksum}
report is reduced (RCCL when the process group is "nccl")."""


def shard_range(total, rank, world):
    """Contiguous block [a, b) of `total` items for `rank` of `world` (remainder to the first ranks)."""
    base, rem = divmod(total, world)
    a = rank * base + min(rank, rem)
    return a, a + base + (1 if rank < rem else 0)


def reduce_report(frames, checksum, device=None):
    """Sum {frames, checksum} over all ranks -> (frames, checksum) everywhere."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(frames), int(checksum)], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t[0]), int(t[1])
