"""SMPL linear blend skinning on the HIP kernels of csrc/smpl_lbs.hip (forward and backward),
exposed as a torch.autograd.Function.  Replaces smplx 0.1.13 ``lbs()`` + the joint handling of
tuch/models/smpl.py:44-56.  No torch-op fallback."""
from __future__ import annotations

import ctypes

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
        with torch.cuda.device(self.device):
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


class _SmplLBS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, betas, pose, dm: SmplDeviceModel, pose2rot: bool):
        L = _C.lib()
        be = betas.detach().to(torch.float32).contiguous()
        po = pose.detach().to(torch.float32).contiguous()
        b = po.shape[0]
        verts = torch.empty(b, dm.num_verts, 3, dtype=torch.float32, device=be.device)
        joints = torch.empty(b, 49, 3, dtype=torch.float32, device=be.device)
        nbytes = L.tuch_smpl_forward_workspace_bytes(dm._handle, b)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=be.device)
        _C.check(L.tuch_smpl_forward(dm._handle, _C.ptr(be), _C.ptr(po), int(pose2rot), b, _C.ptr(verts),
                                     _C.ptr(joints), _C.ptr(ws), nbytes, _C.stream()))
        ctx.dm, ctx.pose2rot, ctx.pose_shape = dm, bool(pose2rot), pose.shape
        ctx.save_for_backward(po, ws)
        return verts, joints

    @staticmethod
    def backward(ctx, g_verts, g_joints):
        L = _C.lib()
        po, ws = ctx.saved_tensors
        b = po.shape[0]
        gv = g_verts.to(torch.float32).contiguous() if g_verts is not None else None
        gj = g_joints.to(torch.float32).contiguous() if g_joints is not None else None
        g_betas = torch.empty(b, 10, dtype=torch.float32, device=po.device)
        g_pose = torch.empty(po.shape, dtype=torch.float32, device=po.device)
        nbytes = L.tuch_smpl_backward_workspace_bytes(ctx.dm._handle, b)
        ws2 = torch.empty(nbytes, dtype=torch.uint8, device=po.device)
        _C.check(L.tuch_smpl_backward(ctx.dm._handle, _C.ptr(po), int(ctx.pose2rot), b, _C.ptr(ws), _C.ptr(gv),
                                      _C.ptr(gj), _C.ptr(g_betas), _C.ptr(g_pose), _C.ptr(ws2), nbytes,
                                      _C.stream()))
        return g_betas, g_pose.view(ctx.pose_shape), None, None


def smpl_forward(smpl_module, betas, full_pose, pose2rot=True):
    """(vertices [B,V,3], joints [B,49,3]) for betas [B,10] and full_pose [B,72] / [B,24,3,3]."""
    dm = getattr(smpl_module, '_device_model', None)
    if dm is None or dm.device != betas.device:
        dm = SmplDeviceModel(smpl_module, betas.device)
        object.__setattr__(smpl_module, '_device_model', dm)
    return _SmplLBS.apply(betas, full_pose, dm, pose2rot)
