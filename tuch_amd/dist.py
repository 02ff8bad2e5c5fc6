"""Data parallelism over bodies.

Every body's loss, gradient, parameters and Adam state are independent (tuch/smplify/losses.py:74,
tuch/train/loss.py:247), so the batch dimension shards across ranks with no data-path collective.
The only exchange is scalar: [sum of losses, body count] for the reported value and -- opt-in,
``RegressorLoss(global_mean=True)`` -- the number of valid bodies for ``contact_loss[valid_fit].mean()``
(tuch/train/loss.py:317) taken over the global batch.  Both are a one- or two-float all-reduce over RCCL
(backend "nccl" on ROCm; gloo in the CPU tests, through a host copy), issued on the calling stream.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


# A group of ONE rank reduces nothing; set True to issue the collective anyway (the one-GPU RCCL smoke of bench.py and
# tests/test_dist.py: init, device all-reduce, capture in a hipGraph, replay, destroy -- on the hardware there is)
REDUCE_SINGLE_RANK = False


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


def all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    """In-place SUM all-reduce of ``t`` over the default group, whatever the backend can take: RCCL ("nccl") reduces the
    device tensor on the calling stream (no host sync, hipGraph-capturable); a backend without device support (gloo in
    the CPU / shared-GPU tests) gets a host copy and the result is copied back (a host round trip: not capturable)."""
    if not (dist.is_available() and dist.is_initialized()):
        return t
    if dist.get_world_size() == 1 and not REDUCE_SINGLE_RANK:
        return t
    backend = str(dist.get_backend()).lower()
    if t.is_cuda and 'nccl' not in backend:
        host = t.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_loss(loss_sum: torch.Tensor, count) -> Tuple[torch.Tensor, torch.Tensor]:
    """(global sum of per-body losses, global body count) as 0-d tensors.  float64 on the wire so
    that the result does not depend on the reduction order at the 1e-6 level."""
    count = count if torch.is_tensor(count) else torch.as_tensor(float(count), device=loss_sum.device)
    stats = torch.stack([loss_sum.detach().to(torch.float64).reshape(()),
                         count.detach().to(torch.float64).reshape(()).to(loss_sum.device)])
    all_reduce_sum(stats)
    return stats[0], stats[1]


def global_count(local_count: torch.Tensor) -> torch.Tensor:
    """Sum of a per-rank count over all ranks (a new tensor; the input is left alone).  No gradient."""
    total = local_count.detach().to(torch.float32).reshape(1).clone()
    all_reduce_sum(total)
    return total[0]


def global_mean_loss(per_body: torch.Tensor, valid: torch.Tensor) -> torch.Tensor:
    """This rank's share of ``per_body[valid].mean()`` taken over ALL ranks' bodies (tuch/train/loss.py:317):
    local sum / global valid count.  Summed over the ranks it is the single-process mean; its gradient with
    respect to this rank's bodies is exactly the single-process gradient -- PROVIDED the ranks' gradients are SUMMED
    (not averaged, as DistributedDataParallel does by default).  Differentiable, no host sync on RCCL."""
    v = valid.to(per_body.dtype)
    return (per_body * v).sum() / global_count(v.sum())
