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


class _DevicePCM:
    """A block of int16 PCM in HBM (sauAmd_Batch_device_pcm) as something torch.as_tensor takes without a copy."""

    def __init__(self, ptr, frames):
        self.__cuda_array_interface__ = {"shape": (int(frames),), "typestr": "<i2", "data": (int(ptr), False), "version": 2}  # (torch takes no read-only flag; nothing here writes)


def device_pcm_tensor(batch, stream, frames, channels=1):
    """The PCM of `stream` as the batch's LAST engine run left it in HBM -> int16 torch tensor [frames * channels] on the
    current device (no copy; valid until the batch renders again or is closed). The PCM block holds one run: a render made
    of several runs has only its last one there (the block is cleared at the start of every run) -- callers that want a
    whole render render it in one run (bench.py --gather-pcm checks that it did)."""
    import torch
    return torch.as_tensor(_DevicePCM(batch.device_pcm(stream), int(frames) * int(channels)), device="cuda")


def gather_renders_to_root(local, dst=0):
    """SURVEY.md 8e, the optional exchange after synthesis: every rank's finished renders -- `local`, an int16 tensor
    [renders, frames], in HBM when the process group is RCCL ("nccl"), in host memory with gloo -- sent straight to
    rank `dst` (torch.distributed.gather is one send per rank to the root inside one group: a direct xGMI hop each,
    no ring). -> on `dst` a tensor [world * renders, frames] in rank order, None elsewhere. Nothing on the data path
    of the renders themselves; ranks must hold equally many renders."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    raw = local.contiguous().view(torch.uint8)  # (neither RCCL nor gloo carries int16: the same bytes as uint8)
    parts = [torch.empty_like(raw) for _ in range(world)] if rank == dst else None
    dist.gather(raw, parts, dst=dst)
    return torch.cat(parts, 0).view(torch.int16) if rank == dst else None
