"""Data parallelism over bodies.

Every body's loss, gradient, parameters and Adam state are independent (tuch/smplify/losses.py:74,
tuch/train/loss.py:247), so the batch dimension shards across ranks with no data-path collective.
The only exchange is the scalar the caller reports: sum of losses and body count (2 floats),
all-reduced over RCCL (backend "nccl" on ROCm) -- or gloo in the CPU tests.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of ``total`` bodies: rank r owns [start, stop); sizes differ by at most one."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_batch(tensor: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_range(tensor.shape[0], rank, world)
    return tensor[lo:hi]


def allreduce_loss(loss_sum: torch.Tensor, count) -> Tuple[torch.Tensor, torch.Tensor]:
    """(global sum of per-body losses, global body count) as 0-d tensors.  float64 on the wire so
    that the result does not depend on the reduction order at the 1e-6 level."""
    stats = torch.stack([loss_sum.detach().to(torch.float64).reshape(()),
                         torch.as_tensor(float(count), dtype=torch.float64, device=loss_sum.device)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    return stats[0], stats[1]


def global_mean_loss(per_body: torch.Tensor, valid: torch.Tensor) -> torch.Tensor:
    """contact_loss[valid_fit].mean() (tuch/train/loss.py:317) over ALL ranks' bodies."""
    v = valid.to(per_body.dtype)
    s, n = allreduce_loss((per_body * v).sum(), float(v.sum().item()))
    return s / n
