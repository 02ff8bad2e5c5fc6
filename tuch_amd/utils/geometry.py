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


def rot6d_to_rotmat(x):
    """6D rotation representation [B,6] (or [B*24*6]) -> [N,3,3] by Gram-Schmidt (reference: geometry.py:67-81)."""
    x = x.view(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = torch.nn.functional.normalize(a1)
    b2 = torch.nn.functional.normalize(a2 - torch.einsum('bi,bi->b', b1, a2).unsqueeze(-1) * b1)
    b3 = torch.cross(b1, b2, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


def estimate_translation(S, joints_2d, focal_length=5000., img_size=224., has_2d_kp_anno=None):
    """Camera translation [B,3] that brings the model joints S [B,49,3] closest to the 2D keypoints
    joints_2d [B,49,3] (u, v, confidence): weighted least squares over the ground-truth joints 25:49 of
    the samples with has_2d_kp_anno, over the OpenPose joints :25 of the others; samples without a
    confident joint get zeros.  Same signature and result as the reference (geometry.py:156-205), which
    loops over the batch on the host with numpy; here one HIP kernel, no host round trip."""
    from .. import _C
    s = S.detach().to(torch.float32).contiguous()
    kp = joints_2d.detach().to(torch.float32).contiguous()
    b, j = s.shape[0], s.shape[1]
    anno = torch.as_tensor(has_2d_kp_anno, device=s.device).to(torch.uint8).contiguous()
    out = torch.empty(b, 3, dtype=torch.float32, device=s.device)
    _C.check(_C.lib().tuch_estimate_translation(_C.ptr(s), _C.ptr(kp), _C.ptr(anno), b, j, float(focal_length),
                                                float(img_size), _C.ptr(out), _C.stream()))
    return out


def estimate_translation_np(S, joints_2d, joints_conf, focal_length=5000, img_size=224):
    """One sample on the host, numpy in / numpy out (reference: geometry.py:114-154; only its own
    estimate_translation calls it, kept for scripts that import the name).  Normal equations of
    sqrt(conf) * (f X + (c - u) t_z ... ) in closed form: rows (f, 0, c-u | (u-c) Z - f X) and the same for v."""
    import numpy as np
    S = np.asarray(S, dtype=np.float64)
    uv = np.asarray(joints_2d, dtype=np.float64)[:, :2] - img_size / 2.
    w = np.sqrt(np.asarray(joints_conf, dtype=np.float64))
    n = S.shape[0]
    Q = np.zeros((n, 2, 3))
    Q[:, 0, 0] = Q[:, 1, 1] = focal_length
    Q[:, :, 2] = -uv
    c = uv * S[:, 2:3] - focal_length * S[:, :2]
    Q = (Q * w[:, None, None]).reshape(2 * n, 3)
    c = (c * w[:, None]).reshape(2 * n)
    return np.linalg.solve(Q.T @ Q, Q.T @ c)


def rotation_matrix_to_angle_axis(rotation_matrix):
    """Rotation matrices [N,3,4] (homogeneous column appended, as the reference's callers do,
    train_module.py:208-211) or [N,3,3] -> angle-axis [N,3]; the torchgeometry 0.1.2 function the
    reference imports.  NaN entries propagate; the callers zero them (:212)."""
    from .. import _C
    r = rotation_matrix.detach().to(torch.float32).contiguous()
    n, stride = r.shape[0], r.shape[2]
    out = torch.empty(n, 3, dtype=torch.float32, device=r.device)
    _C.check(_C.lib().tuch_rotmat_to_angle_axis(_C.ptr(r), n, stride, _C.ptr(out), _C.stream()))
    return out


def angle_axis_to_rotation_matrix(angle_axis):
    """Angle-axis [N,3] -> homogeneous rotation matrices [N,4,4] (torchgeometry 0.1.2 signature, used by the
    reference's fits_dict.py:104): Rodrigues formula, first-order Taylor form below theta^2 = 1e-6."""
    aa = angle_axis
    theta2 = (aa * aa).sum(dim=1)
    theta = torch.sqrt(theta2)
    k = aa / (theta + 1e-6).unsqueeze(1)
    kx, ky, kz = k[:, 0], k[:, 1], k[:, 2]
    c, s = torch.cos(theta), torch.sin(theta)
    one = torch.ones_like(c)
    normal = torch.stack([c + kx * kx * (one - c), kx * ky * (one - c) - kz * s, ky * s + kx * kz * (one - c),
                          kz * s + kx * ky * (one - c), c + ky * ky * (one - c), -kx * s + ky * kz * (one - c),
                          -ky * s + kx * kz * (one - c), kx * s + ky * kz * (one - c), c + kz * kz * (one - c)], dim=1)
    rx, ry, rz = aa[:, 0], aa[:, 1], aa[:, 2]
    taylor = torch.stack([one, -rz, ry, rz, one, -rx, -ry, rx, one], dim=1)
    rot = torch.where((theta2 > 1e-6).unsqueeze(1), normal, taylor).view(-1, 3, 3)
    out = torch.eye(4, dtype=aa.dtype, device=aa.device).repeat(aa.shape[0], 1, 1)
    out[:, :3, :3] = rot
    return out
