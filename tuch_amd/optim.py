"""Adam for the handful of small tensors an SMPLify-DC loop optimises, as ONE kernel launch (csrc/adam.hip).

The reference uses ``torch.optim.Adam`` (tuch/smplify/smplifydc.py:117,150); its capturable implementations take two to
three launches per step, at the very end of every iteration's serial chain.  Same update rule (no weight decay, no
amsgrad), step counter on the device, so a step is capturable in a hipGraph.  Parameters must be float32 HIP tensors of
at most 65536 elements in total, at most 8 of them -- anything else: use torch.optim.Adam.
"""
from __future__ import annotations

import ctypes
import weakref

import numpy as np
import torch

from . import _C

# optimisers created with fuse_backward=True, by the id of each of their parameters (lbs._SmplLBS.backward asks)
_FUSABLE = weakref.WeakValueDictionary()


def fusable_for(global_orient, body_pose):
    """The Adam (created with fuse_backward=True) whose parameters are EXACTLY these two tensor objects, else None."""
    if global_orient is None or body_pose is None:
        return None
    opt = _FUSABLE.get(id(global_orient))
    if opt is None or opt is not _FUSABLE.get(id(body_pose)) or len(opt.params) != 2:
        return None
    if not any(p is global_orient for p in opt.params) or not any(p is body_pose for p in opt.params):
        return None
    return opt


class Adam:
    """Drop-in for the subset of torch.optim.Adam the fitting loops use: ``Adam(params, lr, betas)``, ``step()``,
    ``zero_grad(set_to_none)``, ``state`` (per parameter: ``exp_avg``, ``exp_avg_sq``; plus the shared device ``step``)."""

    MAX_TENSORS, MAX_ELEMENTS = 8, 65536

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, fuse_backward=False, **other):
        # only torch.optim.Adam's plain update: anything that would change it (weight_decay, amsgrad, maximize, ...) is
        # refused, so that make_adam falls back to torch.optim.Adam instead of silently ignoring the setting
        changed = {k: v for k, v in other.items()
                   if not (k in ('weight_decay', 'amsgrad', 'maximize') and not v or k in ('capturable', 'foreach', 'fused',
                                                                                         'differentiable') and v in (None, False))}
        if changed:
            raise ValueError('tuch_amd.optim.Adam supports lr / betas / eps only (got %s); use torch.optim.Adam'
                             % ', '.join(sorted(changed)))
        self.params = list(params)
        if not 0 < len(self.params) <= self.MAX_TENSORS or sum(p.numel() for p in self.params) > self.MAX_ELEMENTS:
            raise ValueError('tuch_amd.optim.Adam is for a few small tensors; use torch.optim.Adam')
        for p in self.params:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise ValueError('tuch_amd.optim.Adam needs contiguous float32 HIP tensors')
        self.lr, self.eps = float(lr), float(eps)
        dev = self.params[0].device
        self.step_count = torch.zeros(1, dtype=torch.float32, device=dev)
        self.state = {p: {'exp_avg': torch.zeros_like(p), 'exp_avg_sq': torch.zeros_like(p), 'step': self.step_count}
                      for p in self.params}
        n = len(self.params)
        self._betas = (ctypes.c_float * (2 * n))(*[float(b) for _ in range(n) for b in betas])
        self._sizes = (ctypes.c_int * n)(*[p.numel() for p in self.params])
        ptrs = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        self._p = ptrs(self.params)
        self._m = ptrs([self.state[p]['exp_avg'] for p in self.params])
        self._v = ptrs([self.state[p]['exp_avg_sq'] for p in self.params])
        self.param_groups = [{'params': self.params, 'lr': self.lr, 'betas': tuple(betas), 'eps': self.eps}]
        # fuse_backward (opt-in, SMPLify-DC stage 2): when the parameters are exactly the body model's two pose tensors and
        # the WHOLE gradient of the objective reaches them through one _SmplLBS node (ops._Stage2Tail is the root of the graph
        # and routes the pose prior's gradient through that node), the node's last backward kernel applies this update itself
        # (tuch_smpl_backward_split_adam) and the following step() is a no-op: one launch less at the end of every iteration.
        # NOTE the consequence: with fuse_backward the parameters move during loss.backward(), not during step().
        self._applied = False
        self._ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        self.fuse_backward = bool(fuse_backward)
        if self.fuse_backward:
            for p in self.params:
                _FUSABLE[id(p)] = self

    def zero_grad(self, set_to_none: bool = True) -> None:
        # a fused update whose step() never came (an exception, a gradient-only evaluation) must not swallow the NEXT step()
        self._applied = False
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def reset(self) -> None:
        """A fresh optimiser (the reference creates one per stage and call): moments and step counter zeroed in place."""
        torch._foreach_zero_([self.step_count] + [s[k] for s in self.state.values() for k in ('exp_avg', 'exp_avg_sq')])
        self._applied = False

    def step(self) -> None:
        if self._applied:               # the backward pass has applied this update already (fuse_backward)
            self._applied = False
            return
        grads = []
        for p in self.params:
            if p.grad is None:
                raise RuntimeError('tuch_amd.optim.Adam.step: a parameter has no gradient')
            grads.append(p.grad if p.grad.is_contiguous() else p.grad.contiguous())
        n = len(self.params)
        g = (ctypes.c_void_p * n)(*[t.data_ptr() for t in grads])
        # lr / eps as they are NOW in param_groups (a scheduler or the caller may have changed them; a captured hipGraph keeps
        # the values it was captured with, like torch's own non-tensor lr)
        group = self.param_groups[0]
        _C.check(_C.lib().tuch_adam_step(n, self._p, g, self._m, self._v, self._sizes, self._betas,
                                         _C.ptr(self.step_count), float(group['lr']), float(group['eps']), _C.stream()))


def make_adam(params, lr, capturable=True, fuse_backward=False, **adam_kwargs):
    """tuch_amd.optim.Adam where it applies (HIP float32 parameters, few and small), else torch.optim.Adam."""
    params = list(params)
    try:
        return Adam(params, lr=lr, fuse_backward=fuse_backward, **adam_kwargs)
    except ValueError:
        return torch.optim.Adam(params, lr=lr, capturable=capturable and params[0].is_cuda, **adam_kwargs)
