"""Event-range sharding over GPUs / processes.

Events are independent and every random draw is keyed by the *global* event id, so a batch
shards as contiguous id ranges with no data-path collective: rank r of W simulates
``[first + r*per_rank, first + (r+1)*per_rank)`` (weak scaling: fixed work per GPU) or an
even split of a fixed total (strong scaling).  The only cross-rank traffic is control
plane: a barrier around the timed region and a MAX / SUM of a few scalars, done with
``torch.distributed`` when it is initialised.
"""
from __future__ import annotations

import os


def world() -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process => 0, 0, 1)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def weak_shard(per_rank: int, rank: int, first_event: int = 0) -> tuple[int, int]:
    """(first_event, n_events) of `rank` when every rank simulates `per_rank` events."""
    return first_event + rank * per_rank, per_rank


def strong_shard(total: int, rank: int, world_size: int, first_event: int = 0) -> tuple[int, int]:
    """(first_event, n_events) of `rank` for an even contiguous split of `total` events."""
    base, extra = divmod(total, world_size)
    start = rank * base + min(rank, extra)
    return first_event + start, base + (1 if rank < extra else 0)


def init_process_group(backend: str = "gloo"):
    """Join the torchrun rendezvous (control plane only).  Returns the dist module or None."""
    rank, _, world_size = world()
    if world_size <= 1:
        return None
    import torch.distributed as dist

    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world_size)
    return dist


def barrier(dist) -> None:
    if dist is not None:
        dist.barrier()


def reduce_scalars(dist, values: list[float], op: str) -> list[float]:
    """MAX or SUM of a few float64 scalars over ranks (identity without a process group)."""
    if dist is None:
        return list(values)
    import torch

    t = torch.tensor(values, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return [float(v) for v in t]


def reduce_checksums(dist, values: list[int]) -> list[int]:
    """Sum of uint64 checksums mod 2^64 over ranks (as two 32-bit halves in int64 tensors)."""
    if dist is None:
        return [v % (1 << 64) for v in values]
    import torch

    halves = []
    for v in values:
        halves += [v & 0xFFFFFFFF, (v >> 32) & 0xFFFFFFFF]
    t = torch.tensor(halves, dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    out = []
    for i in range(len(values)):
        out.append((int(t[2 * i]) + (int(t[2 * i + 1]) << 32)) % (1 << 64))
    return out
