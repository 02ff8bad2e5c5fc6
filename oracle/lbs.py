"""CPU oracle for the SMPL forward pass -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference delegates this arithmetic to the third-party
package smplx==0.1.13 (requirements.txt:13; call sites tuch/models/smpl.py:22,24,46,47),
which is neither vendored under /root/reference nor installable here, and the
reference has no test that pins it.  This file restates the published algorithm
of smplx/lbs.py (lbs, blend_shapes, vertices2joints, batch_rodrigues,
transform_mat, batch_rigid_transform) and smplx/body_models.py (SMPL.forward,
VertexJointSelector) as summarised in SURVEY.md §3.3, plus the reference's own
wrapper tuch/models/smpl.py:44-56 (9 extra regressed joints, 49-entry joint map).
It is checked by mathematical known-answer tests (tests/test_oracle_lbs.py): zero
pose, root-only rotation = rigid rotation, fp64 agreement, finite differences.

Written on torch CPU tensors so that autograd provides the backward oracle.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch


def model_tensors(body, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """SyntheticBody -> the buffers smplx.SMPL registers, as torch CPU tensors."""
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
    return dict(
        v_template=t(body.v_template), shapedirs=t(body.shapedirs), posedirs=t(body.posedirs),
        J_regressor=t(body.J_regressor), lbs_weights=t(body.lbs_weights),
        parents=torch.as_tensor(body.parents, dtype=torch.long),
        extra_vertex_ids=torch.as_tensor(body.extra_vertex_ids, dtype=torch.long),
        J_regressor_extra=t(body.J_regressor_extra),
        joint_map=torch.as_tensor(body.joint_map, dtype=torch.long))


def rodrigues(aa: torch.Tensor) -> torch.Tensor:
    """smplx batch_rodrigues: angle = |aa + 1e-8|, R = I + sin K + (1 - cos) K K.  [N,3] -> [N,3,3]."""
    angle = torch.linalg.vector_norm(aa + 1e-8, dim=1, keepdim=True)
    axis = aa / angle
    zero = torch.zeros_like(axis[:, 0])
    x, y, z = axis[:, 0], axis[:, 1], axis[:, 2]
    skew = torch.stack([zero, -z, y, z, zero, -x, -y, x, zero], 1).reshape(-1, 3, 3)
    s = torch.sin(angle)[:, :, None]
    c = torch.cos(angle)[:, :, None]
    eye = torch.eye(3, dtype=aa.dtype).expand_as(skew)
    return eye + s * skew + (1.0 - c) * (skew @ skew)


def rigid_chain(rot: torch.Tensor, joints: torch.Tensor, parents: torch.Tensor
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """smplx batch_rigid_transform: world transforms along the tree and the
    transforms relative to the rest pose.  rot [B,J,3,3], joints [B,J,3]."""
    bsz, nj = joints.shape[:2]
    rel = joints.clone()
    rel[:, 1:] = joints[:, 1:] - joints[:, parents[1:]]
    local = torch.zeros(bsz, nj, 4, 4, dtype=rot.dtype)
    local[:, :, :3, :3] = rot
    local[:, :, :3, 3] = rel
    local[:, :, 3, 3] = 1.0
    world = [local[:, 0]]
    for k in range(1, nj):
        world.append(world[int(parents[k])] @ local[:, k])
    world = torch.stack(world, 1)
    posed = world[:, :, :3, 3]
    # subtract the rest-pose joint carried through the transform (translation column only)
    carried = (world[:, :, :3, :3] @ joints[..., None])[..., 0]
    rel_world = world.clone()
    rel_world[:, :, :3, 3] = world[:, :, :3, 3] - carried
    return posed, rel_world


def lbs(betas, full_pose, m, pose2rot=True):
    """smplx lbs(): returns (verts [B,V,3], posed joints [B,24,3])."""
    bsz = betas.shape[0]
    v_shaped = m['v_template'][None] + torch.einsum('bl,vkl->bvk', betas, m['shapedirs'])
    joints = torch.einsum('bvk,jv->bjk', v_shaped, m['J_regressor'])
    if pose2rot:
        rot = rodrigues(full_pose.reshape(-1, 3)).reshape(bsz, -1, 3, 3)
    else:
        rot = full_pose.reshape(bsz, -1, 3, 3)
    feat = (rot[:, 1:] - torch.eye(3, dtype=rot.dtype)).reshape(bsz, -1)
    v_posed = v_shaped + (feat @ m['posedirs']).reshape(bsz, -1, 3)
    posed_joints, rel = rigid_chain(rot, joints, m['parents'])
    blend = (m['lbs_weights'] @ rel.reshape(bsz, -1, 16)).reshape(bsz, -1, 4, 4)
    verts = (blend[:, :, :3, :3] @ v_posed[..., None])[..., 0] + blend[:, :, :3, 3]
    return verts, posed_joints


def smpl_forward(m, betas, body_pose, global_orient, pose2rot=True):
    """tuch/models/smpl.py:44-56 on top of smplx SMPL.forward: (vertices, joints[49])."""
    full = torch.cat([global_orient, body_pose], 1)
    verts, joints = lbs(betas, full, m, pose2rot)
    picked = verts[:, m['extra_vertex_ids']]                       # VertexJointSelector
    extra = torch.einsum('bvk,jv->bjk', verts, m['J_regressor_extra'])   # smpl.py:47
    joints = torch.cat([joints, picked, extra], 1)[:, m['joint_map']]    # smpl.py:48-49
    return verts, joints
