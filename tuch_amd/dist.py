"""Data parallelism over bodies.

Every body's loss, gradient, parameters and Adam state are independent (tuch/smplify/losses.py:74,
tuch/train/loss.py:247), so the batch dimension shards across ranks with no data-path collective.
The only exchange is scalar: [sum of losses, body count] for the reported value, and the number of
valid bodies for ``contact_loss[valid_fit].mean()`` (tuch/train/loss.py:317, a mean over ALL valid
bodies of the global batch).  Both are a one- or two-float all-reduce over RCCL (backend "nccl" on
ROCm; gloo in the CPU tests), issued on the calling stream, with no host synchronisation.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


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
    count = count if torch.is_tensor(count) else torch.as_tensor(float(count), device=loss_sum.device)
    stats = torch.stack([loss_sum.detach().to(torch.float64).reshape(()),
                         count.detach().to(torch.float64).reshape(()).to(loss_sum.device)])
    if world_size() > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    return stats[0], stats[1]


def global_count(local_count: torch.Tensor) -> torch.Tensor:
    """Sum of a per-rank count over all ranks (a new tensor; the input is left alone).  No gradient."""
    total = local_count.detach().to(torch.float32).reshape(1).clone()
    if world_size() > 1:
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
    return total[0]


def global_mean_loss(per_body: torch.Tensor, valid: torch.Tensor) -> torch.Tensor:
    """This rank's share of ``per_body[valid].mean()`` taken over ALL ranks' bodies (tuch/train/loss.py:317):
    local sum / global valid count.  Summed over the ranks it is the single-process mean; its gradient with
    respect to this rank's bodies is exactly the single-process gradient.  Differentiable, no host sync."""
    v = valid.to(per_body.dtype)
    return (per_body * v).sum() / global_count(v.sum())
