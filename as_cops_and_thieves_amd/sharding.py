"""Env-slot sharding across the GPUs of one node.

Env instances are independent (one ``pymunk.Space`` each in the reference, base_env.py:77), so
the simulator needs NO data-path collective: rank r owns global env ids
``[offset_r, offset_r + n_r)`` and seeds its Philox streams with those ids.  The only exchange is
the timing reduction of the benchmark (and, for a learner, its gradient all-reduce).
"""
from __future__ import annotations

from typing import Tuple


def shard_envs(total_envs: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous partition of ``total_envs`` global env ids -> ``(n_local, offset)`` of ``rank``."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(total_envs, world_size)
    n_local = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return n_local, offset


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a python float (identity when torch.distributed is not initialised)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
