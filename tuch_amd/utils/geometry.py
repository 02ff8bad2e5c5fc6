"""The two geometry helpers on the SMPLify-DC inner loop (reference: tuch/utils/geometry.py).
Tiny [B,49,*] tensors: plain torch ops on the inputs' device (K8 of SURVEY.md §2.2)."""
from __future__ import annotations

import torch


def batch_rodrigues(theta):
    """Axis-angle [N,3] -> rotation matrices [N,3,3] via the unit quaternion
    (reference: geometry.py:29-43, note the 1e-8 added before the norm)."""
    angle = torch.norm(theta + 1e-8, p=2, dim=1, keepdim=True)
    axis = theta / angle
    half = 0.5 * angle
    return quat_to_rotmat(torch.cat([torch.cos(half), torch.sin(half) * axis], dim=1))


def quat_to_rotmat(quat):
    """(w,x,y,z) [N,4] -> [N,3,3] (reference: geometry.py:45-65)."""
    q = quat / quat.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    ww, xx, yy, zz = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    rows = [ww + xx - yy - zz, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
            2 * wz + 2 * xy, ww - xx + yy - zz, 2 * yz - 2 * wx,
            2 * xz - 2 * wy, 2 * wx + 2 * yz, ww - xx - yy + zz]
    return torch.stack(rows, dim=1).view(-1, 3, 3)


def perspective_projection(points, rotation, translation, focal_length, camera_center):
    """Pinhole projection (reference: geometry.py:83-111): x' = R x + t, uv = f * x'_xy / x'_z + c."""
    cam = torch.einsum('bij,bkj->bki', rotation, points) + translation.unsqueeze(1)
    ndc = cam[:, :, :2] / cam[:, :, 2:3]
    if torch.is_tensor(focal_length) and focal_length.dim() > 0:
        focal_length = focal_length.view(-1, 1, 1)
    return focal_length * ndc + camera_center.unsqueeze(1)
