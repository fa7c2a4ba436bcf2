"""Image sharding over ranks: images are independent units, so a job over N GPUs is N
replicas of the single-GPU path, each on its own contiguous shard; there is no collective on the data
path.  The only cross-rank steps are a barrier around the timed region and the MAX of the elapsed times."""
from __future__ import annotations

from typing import Tuple


def shard(n_units: int, rank: int, world: int) -> Tuple[int, int]:
    """Half-open range [lo, hi) of `n_units` independent units owned by `rank`; sizes differ by at most one."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} of world {world}")
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(seconds: float, device=None) -> float:
    """MAX of a per-rank duration over the process group (identity when not distributed)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
