"""SMPL linear blend skinning on the HIP kernels of csrc/smpl_lbs.hip (forward and backward),
exposed as a torch.autograd.Function.  Replaces smplx 0.1.13 ``lbs()`` + the joint handling of
tuch/models/smpl.py:44-56.  No torch-op fallback."""
from __future__ import annotations

import ctypes
import weakref

import numpy as np
import torch

from . import _C


class SmplDeviceModel:
    """Device copy of one SMPL model's constants (wraps tuch_smpl_model)."""

    def __init__(self, smpl_module, device):
        g = lambda name, dt: np.ascontiguousarray(getattr(smpl_module, name).detach().cpu().numpy().astype(dt))
        self.device = torch.device(device)
        self.num_verts = int(smpl_module.v_template.shape[0])
        arrays = [g('v_template', np.float32), g('shapedirs', np.float32), g('posedirs', np.float32),
                  g('J_regressor', np.float32), g('lbs_weights', np.float32), g('parents', np.int32),
                  g('extra_vertex_ids', np.int32), g('J_regressor_extra', np.float32),
                  np.ascontiguousarray(smpl_module.joint_map.cpu().numpy().astype(np.int32))]
        assert arrays[1].shape == (self.num_verts, 3, 10) and arrays[2].shape == (207, 3 * self.num_verts)
        assert arrays[3].shape == (24, self.num_verts) and arrays[4].shape == (self.num_verts, 24)
        assert arrays[6].shape == (21,) and arrays[7].shape == (9, self.num_verts) and arrays[8].shape == (49,)
        handle = ctypes.c_void_p(0)
        import contextlib
        import os
        host_tables = os.environ.get('TUCH_HOST_TABLES', '0') not in ('', '0')     # sanitizer runs of the table builders
        with (contextlib.nullcontext() if host_tables else torch.cuda.device(self.device)):
            _C.check(_C.lib().tuch_smpl_model_create(ctypes.byref(handle), self.num_verts,
                                                     *[a.ctypes.data_as(ctypes.c_void_p) for a in arrays]))
        self._handle = handle

    def __del__(self):
        h = getattr(self, '_handle', None)
        if h:
            try:
                _C.lib().tuch_smpl_model_destroy(h)
            except Exception:
                pass
            self._handle = None


def _pose_rows(t, n):
    """[B,n] float32 rows of a pose tensor; row views of one concatenated [B,72] pose pass as they are (row stride)."""
    r = t.detach().to(torch.float32).reshape(-1, n)
    return r if r.stride(1) == 1 and r.stride(0) >= n else r.contiguous()


class _SmplLBS(torch.autograd.Function):
    """betas [B,10], global_orient [B,3] / [B,1,3,3], body_pose [B,69] / [B,23,3,3] -> vertices, joints.  The two pose
    tensors go to the kernels as they are (no concatenated copy, and the gradient comes back as two tensors)."""

    @staticmethod
    def forward(ctx, betas, global_orient, body_pose, dm: SmplDeviceModel, pose2rot: bool, full=None):
        L = _C.lib()
        w = 3 if pose2rot else 9
        be = betas.detach().to(torch.float32).contiguous()
        # full: ONE float32 tensor [B, 24 w] that global_orient and body_pose are the row views [:, :w] and [:, w:] of
        # (smpl_forward_split found them to be): the gradient then goes back to it as one tensor -- autograd does not
        # have to zero-fill and copy two slice gradients and add them (five launches per step of a train.py-style loop,
        # which passes pred_rotmat[:, 1:] and pred_rotmat[:, 0].unsqueeze(1), train_module.py:202-204)
        ctx.full = full is not None
        ctx.set_materialize_grads(False)        # an output nobody differentiates arrives as None, not as a zero-filled tensor
        go, bp = _pose_rows(global_orient, w), _pose_rows(body_pose, 23 * w)
        b = go.shape[0]
        if bp.shape[0] != b or be.shape[0] != b:
            raise ValueError(f'SMPL: batch sizes differ (betas {be.shape[0]}, global_orient {b}, body_pose {bp.shape[0]})')
        verts = torch.empty(b, dm.num_verts, 3, dtype=torch.float32, device=be.device)
        joints = torch.empty(b, 49, 3, dtype=torch.float32, device=be.device)
        nbytes = L.tuch_smpl_forward_workspace_bytes(dm._handle, b)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=be.device)
        _C.check(L.tuch_smpl_forward_split(dm._handle, _C.ptr(be), _C.row_ptr(go), go.stride(0), _C.row_ptr(bp), bp.stride(0),
                                           int(pose2rot), b, _C.ptr(verts), _C.ptr(joints), _C.ptr(ws), nbytes, _C.stream()))
        ctx.dm, ctx.pose2rot = dm, bool(pose2rot)
        ctx.shapes = (global_orient.shape, body_pose.shape)
        # A later node that holds ANOTHER gradient for the same body_pose (ops._Stage2Tail: the pose prior) may leave it in
        # pose_grad_extra instead of returning it: backward() then adds it inside its last kernel and autograd has nothing
        # left to sum (one add launch less per step).  pose_ref: how that node recognises the tensor -- by IDENTITY (the very
        # tensor object this node will return a gradient for; an alias with the same address and shape is another leaf).
        ctx.pose_ref = weakref.ref(body_pose) if ctx.needs_input_grad[2] else None
        ctx.orient_ref = weakref.ref(global_orient) if ctx.needs_input_grad[1] else None
        ctx.pose_grad_extra = None
        ctx.verts_grad_fixed = None   # set by ops._Stage2Tail.backward (deterministic mode): (fixed-point vertex gradient, pass id)
        ctx.root_pass = None          # set by ops._Stage2Tail.backward: the id of a backward pass whose ROOT is that node
        ctx.save_for_backward(go, bp, ws)
        return verts, joints

    @staticmethod
    def backward(ctx, g_verts, g_joints):
        L = _C.lib()
        go, bp, ws = ctx.saved_tensors
        b = go.shape[0]
        w = go.shape[1]
        if g_verts is None and g_joints is None:
            return None, None, None, None, None, None
        gv = g_verts.to(torch.float32).contiguous() if g_verts is not None else None
        gj = g_joints.to(torch.float32).contiguous() if g_joints is not None else None
        g_betas = torch.empty(b, 10, dtype=torch.float32, device=go.device)
        nbytes = L.tuch_smpl_backward_workspace_bytes(ctx.dm._handle, b)
        ws2 = torch.empty(nbytes, dtype=torch.uint8, device=go.device)
        if ctx.full:
            # the kernel writes the two gradients with a row stride: one [B, 24 w] tensor takes both
            g_full = torch.empty(b, 24 * w, dtype=torch.float32, device=go.device)
            _C.check(L.tuch_smpl_backward_split_add(ctx.dm._handle, _C.row_ptr(go), go.stride(0), _C.row_ptr(bp), bp.stride(0),
                                                    int(ctx.pose2rot), b, _C.ptr(ws), _C.ptr(gv), _C.ptr(gj),
                                                    _C.ptr(g_betas), _C.row_ptr(g_full[:, :w]), 24 * w,
                                                    _C.row_ptr(g_full[:, w:]), 24 * w, None, 23 * w,
                                                    _C.ptr(ws2), nbytes, _C.stream(), None))
            return g_betas, None, None, None, None, g_full
        g_go = torch.empty(go.shape, dtype=torch.float32, device=go.device)
        g_bp = torch.empty(bp.shape, dtype=torch.float32, device=go.device)
        tagged, ctx.pose_grad_extra = ctx.pose_grad_extra, None
        extra = None
        f = getattr(torch._C, '_current_graph_task_id', None)
        task = int(f()) if f is not None else -1
        if tagged is not None and task >= 0 and task == tagged[1]:          # left by a node of THIS backward pass
            extra = tagged[0].to(torch.float32).reshape(b, 23 * w).contiguous()
        # the stage-2 tail's vertex gradient as 64-bit fixed-point sums (deterministic mode, ops._Stage2Tail): read by the
        # skinning adjoint itself, added to g_verts (zeros from that node, plus whatever else flows into the vertices)
        tagged_fixed, ctx.verts_grad_fixed = ctx.verts_grad_fixed, None
        fixed = tagged_fixed[0] if tagged_fixed is not None and task >= 0 and task == tagged_fixed[1] else None
        # Adam inside the last backward kernel (optim.Adam(fuse_backward=True)): only when this pass's root is the stage-2
        # objective node, which has routed the prior's gradient here -- then what this call computes IS the whole gradient
        # of the two pose tensors
        adam = None
        root_pass, ctx.root_pass = ctx.root_pass, None
        if ctx.pose2rot and extra is not None and task >= 0 and root_pass == task and ctx.orient_ref is not None \
                and not ctx.needs_input_grad[0]:
            from . import optim
            go_t, bp_t = ctx.orient_ref(), ctx.pose_ref() if ctx.pose_ref is not None else None
            adam = optim.fusable_for(go_t, bp_t)
            if adam is not None and not (go_t.is_contiguous() and bp_t.is_contiguous() and go_t.shape == (b, 3)
                                         and bp_t.shape == (b, 69) and go_t.data_ptr() == go.data_ptr()
                                         and bp_t.data_ptr() == bp.data_ptr()):
                adam = None
        if adam is not None and adam._applied:
            # a second backward pass before step() / zero_grad(): the first pass has already moved the parameters; applying
            # another update here would be a second optimiser step nobody asked for -- this pass only returns gradients.
            # (Gradient ACCUMULATION over several backward passes is not what fuse_backward can do: the first pass has stepped
            # with its own gradient alone; torch.optim.Adam would step once with the sum.)
            import warnings
            warnings.warn('tuch_amd.optim.Adam(fuse_backward=True): a second backward pass before step() -- the parameters '
                          'were already updated with the first pass\'s gradient; use fuse_backward=False to accumulate '
                          'gradients over several passes', RuntimeWarning, stacklevel=2)
            adam = None
        if adam is not None:
            group = adam.param_groups[0]
            st_go, st_bp = adam.state[go_t], adam.state[bp_t]
            _C.check(L.tuch_smpl_backward_split_adam(
                ctx.dm._handle, _C.row_ptr(go), go.stride(0), _C.row_ptr(bp), bp.stride(0), b, _C.ptr(ws), _C.ptr(gv), _C.ptr(gj),
                _C.ptr(g_betas), _C.ptr(g_go), w, _C.ptr(g_bp), 23 * w, _C.ptr(extra), 23 * w,
                _C.ptr(go_t), go_t.stride(0), _C.ptr(bp_t), bp_t.stride(0), _C.ptr(st_go['exp_avg']), _C.ptr(st_go['exp_avg_sq']),
                _C.ptr(st_bp['exp_avg']), _C.ptr(st_bp['exp_avg_sq']), _C.ptr(adam.step_count), _C.ptr(adam._ticket),
                float(group['lr']), float(group['betas'][0]), float(group['betas'][1]), float(group['eps']),
                _C.ptr(ws2), nbytes, _C.stream(), _C.ptr(fixed)))
            adam._applied = True
            return g_betas, g_go.view(ctx.shapes[0]), g_bp.view(ctx.shapes[1]), None, None, None
        _C.check(L.tuch_smpl_backward_split_add(ctx.dm._handle, _C.row_ptr(go), go.stride(0), _C.row_ptr(bp), bp.stride(0),
                                                int(ctx.pose2rot), b, _C.ptr(ws), _C.ptr(gv), _C.ptr(gj),
                                                _C.ptr(g_betas), _C.ptr(g_go), w, _C.ptr(g_bp), 23 * w,
                                                _C.ptr(extra), 23 * w, _C.ptr(ws2), nbytes, _C.stream(), _C.ptr(fixed)))
        return g_betas, g_go.view(ctx.shapes[0]), g_bp.view(ctx.shapes[1]), None, None, None


def _device_model(smpl_module, device):
    dm = getattr(smpl_module, '_device_model', None)
    if dm is None or dm.device != device:
        dm = SmplDeviceModel(smpl_module, device)
        object.__setattr__(smpl_module, '_device_model', dm)
    return dm


def _shared_rows(global_orient, body_pose, w):
    """The [B, 24 w] float32 tensor the two pose tensors are the row views [:, :w] / [:, w:] of, or None: both views of ONE
    contiguous base holding exactly the B x 24 w values, the orientation first."""
    base = getattr(global_orient, '_base', None)
    if base is None or getattr(body_pose, '_base', None) is not base or base.dtype != torch.float32:
        return None
    if not base.is_contiguous() or base.dim() < 2 or global_orient.dim() < 2 or body_pose.dim() < 2:
        return None
    b = base.shape[0]
    if base.numel() != b * 24 * w or global_orient.shape[0] != b or body_pose.shape[0] != b:
        return None
    if global_orient.numel() != b * w or body_pose.numel() != b * 23 * w:
        return None
    try:
        go, bp = global_orient.view(b, w), body_pose.view(b, 23 * w)       # (raises if not expressible as row views)
    except RuntimeError:
        return None
    if go.stride() != (24 * w, 1) or bp.stride() != (24 * w, 1):
        return None
    if go.storage_offset() != base.storage_offset() or bp.storage_offset() != base.storage_offset() + w:
        return None
    return base.view(b, 24 * w)


def _plain_view(t) -> bool:
    return t.grad_fn is not None and not t.retains_grad and not getattr(t, '_backward_hooks', None)


def smpl_forward_split(smpl_module, betas, global_orient, body_pose, pose2rot=True):
    """(vertices [B,V,3], joints [B,49,3]) from the two pose tensors of SMPL.forward (tuch/models/smpl.py:44-47)."""
    dm = _device_model(smpl_module, betas.device)
    if torch.is_grad_enabled() and (global_orient.requires_grad or body_pose.requires_grad):
        full = _shared_rows(global_orient, body_pose, 3 if pose2rot else 9)
        # one gradient to the shared base only when that is where autograd would send the two gradients anyway: the base
        # is differentiable and both views hang off it (a leaf view of a non-differentiable base made a parameter with
        # requires_grad_() keeps its own .grad), and nobody watches the views themselves (hooks, retain_grad)
        if full is not None and not (full.requires_grad and _plain_view(global_orient) and _plain_view(body_pose)):
            full = None
        if full is not None:
            return _SmplLBS.apply(betas, global_orient.detach(), body_pose.detach(), dm, pose2rot, full)
    return _SmplLBS.apply(betas, global_orient, body_pose, dm, pose2rot)


def smpl_forward(smpl_module, betas, full_pose, pose2rot=True):
    """(vertices [B,V,3], joints [B,49,3]) for betas [B,10] and full_pose [B,72] / [B,24,3,3]."""
    w = 3 if pose2rot else 9
    flat = full_pose.reshape(full_pose.shape[0], 24 * w)
    return smpl_forward_split(smpl_module, betas, flat[:, :w], flat[:, w:], pose2rot)
