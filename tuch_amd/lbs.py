"""SMPL linear blend skinning entry point used by tuch_amd.models.smpl.SMPL.

TEMPORARY (round 1, being replaced by the HIP kernels in csrc/smpl_lbs.hip): device-side torch
ops with autograd.  Algorithm: smplx 0.1.13 lbs() + tuch/models/smpl.py:44-56.
"""
from __future__ import annotations

import torch


def _rodrigues(aa):
    angle = torch.norm(aa + 1e-8, dim=1, keepdim=True)
    d = aa / angle
    c, s = torch.cos(angle)[:, :, None], torch.sin(angle)[:, :, None]
    z = torch.zeros_like(d[:, 0])
    k = torch.stack([z, -d[:, 2], d[:, 1], d[:, 2], z, -d[:, 0], -d[:, 1], d[:, 0], z], 1).view(-1, 3, 3)
    return torch.eye(3, dtype=aa.dtype, device=aa.device)[None] + s * k + (1 - c) * torch.bmm(k, k)


def smpl_forward(m, betas, full_pose, pose2rot=True):
    bsz = full_pose.shape[0]
    dev, dt = betas.device, betas.dtype
    v_shaped = m.v_template[None] + torch.einsum('bl,vkl->bvk', betas, m.shapedirs)
    joints = torch.einsum('bvk,jv->bjk', v_shaped, m.J_regressor)
    rot = _rodrigues(full_pose.reshape(-1, 3)).view(bsz, 24, 3, 3) if pose2rot else full_pose.reshape(bsz, 24, 3, 3)
    feat = (rot[:, 1:] - torch.eye(3, dtype=dt, device=dev)).reshape(bsz, 207)
    v_posed = v_shaped + torch.matmul(feat, m.posedirs).view(bsz, -1, 3)
    parents = m.parents.tolist()
    rel = torch.cat([joints[:, :1], joints[:, 1:] - joints[:, parents[1:]]], 1)
    world_r, world_t = [rot[:, 0]], [rel[:, 0]]
    for k in range(1, 24):
        p = parents[k]
        world_r.append(torch.bmm(world_r[p], rot[:, k]))
        world_t.append(torch.bmm(world_r[p], rel[:, k, :, None])[..., 0] + world_t[p])
    wr, wt = torch.stack(world_r, 1), torch.stack(world_t, 1)
    rel_t = wt - torch.matmul(wr, joints[..., None])[..., 0]
    a = torch.cat([wr, rel_t[..., None]], -1).reshape(bsz, 24, 12)
    t = torch.matmul(m.lbs_weights, a).view(bsz, -1, 3, 4)
    verts = torch.matmul(t[..., :3], v_posed[..., None])[..., 0] + t[..., 3]
    picked = verts[:, m.extra_vertex_ids]
    extra = torch.einsum('bvk,jv->bjk', verts, m.J_regressor_extra)
    all_joints = torch.cat([wt, picked, extra], 1)[:, m.joint_map.to(dev)]
    return verts, all_joints
