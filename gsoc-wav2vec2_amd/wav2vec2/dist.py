"""Data-parallel helpers: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests).

The forward path shards along batch only -- utterances are independent (GroupNorm / LayerNorm
statistics are per sample, attention is within an utterance, CTC is per sample), which is exactly the
reference's scheme (src/main.py:156,192).  No collective touches the data path of a forward; the only
cross-rank traffic is the timing reduction and an optional gather of the outputs.
"""

import os


def env_world():
    """(world_size, rank, local_rank) as torch.distributed.run exports them."""
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend="nccl", device=None):
    """Initialise the default process group from the launcher's env (MASTER_ADDR defaults to 127.0.0.1)."""
    import torch.distributed as dist
    world, rank, _ = env_world()
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ      # under torch.distributed.run, even with 1 rank
    if dist.is_initialized() or (world == 1 and not launched):
        return world, rank
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return world, rank


def shard_bounds(total_rows, world, rank):
    """Contiguous [lo, hi) slice of a global batch for `rank`; the first total % world ranks take one
    extra row (the reference keeps a fixed 32 rows per replica, src/main.py:41,156)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    q, r = divmod(int(total_rows), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def barrier(sync_device=None):
    import torch.distributed as dist
    if sync_device is not None:
        sync_device()
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if sync_device is not None:
        sync_device()


def max_over_ranks(value, device="cpu"):
    """MAX all-reduce of a python float (the slowest rank defines the step time)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows(local, total_rows, device=None):
    """All-gather row shards (possibly ragged) back into the full (total_rows, ...) tensor, in rank order."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size()
    sizes = [shard_bounds(total_rows, world, r) for r in range(world)]
    max_rows = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)
