"""CPU: known-answer tests of the SMPL LBS oracle.  The reference delegates this arithmetic to
smplx==0.1.13, which is absent (parity unpinned, see oracle/lbs.py); the oracle is therefore
checked against mathematics, not against reference outputs."""
import numpy as np
import pytest
import torch

from helpers import assert_close
from oracle import lbs as ol
from synthetic import make_body, random_poses


@pytest.fixture(scope='module')
def body():
    return make_body(12, 14, with_geodesics=False)


def test_zero_pose_is_shape_blend_only(body):
    m = ol.model_tensors(body)
    betas = torch.tensor(random_poses(2, 3)[2])
    v, j = ol.lbs(betas, torch.zeros(2, 72), m)
    want = m['v_template'][None] + torch.einsum('bl,vkl->bvk', betas, m['shapedirs'])
    assert_close(v.numpy(), want.numpy(), 0, 2e-6, 'zero-pose verts')
    assert_close(j.numpy(), torch.einsum('bvk,jv->bjk', want, m['J_regressor']).numpy(), 0, 2e-6, 'joints')


def test_root_rotation_is_rigid(body):
    m = ol.model_tensors(body, torch.float64)
    betas = torch.tensor(random_poses(1, 4)[2], dtype=torch.float64)
    pose = torch.zeros(1, 72, dtype=torch.float64)
    v0, j0 = ol.lbs(betas, pose, m)
    pose[0, :3] = torch.tensor([0.3, -0.5, 0.2])
    v1, j1 = ol.lbs(betas, pose, m)
    rot = ol.rodrigues(pose[:, :3])[0]
    root = j0[0, 0]
    assert_close(v1[0].numpy(), ((v0[0] - root) @ rot.T + root).numpy(), 0, 1e-9, 'rigid verts')
    assert_close(torch.det(rot).item(), 1.0, 0, 1e-12, 'det')


def test_float32_tracks_float64(body):
    bp, go, be = random_poses(3, 7)
    out32 = ol.smpl_forward(ol.model_tensors(body), torch.tensor(be), torch.tensor(bp), torch.tensor(go))
    t64 = lambda a: torch.tensor(a, dtype=torch.float64)
    out64 = ol.smpl_forward(ol.model_tensors(body, torch.float64), t64(be), t64(bp), t64(go))
    assert_close(out32[0].numpy(), out64[0].numpy(), 1e-5, 2e-6, 'verts')
    assert_close(out32[1].numpy(), out64[1].numpy(), 1e-5, 2e-6, 'joints')
    assert out32[1].shape == (3, 49, 3)


def test_rotmat_input_equals_axis_angle_input(body):
    m = ol.model_tensors(body, torch.float64)
    bp, go, be = [torch.tensor(a, dtype=torch.float64) for a in random_poses(2, 9)]
    full = torch.cat([go, bp], 1)
    va, ja = ol.lbs(be, full, m, pose2rot=True)
    rot = ol.rodrigues(full.reshape(-1, 3)).reshape(2, 24, 3, 3)
    vr, jr = ol.lbs(be, rot, m, pose2rot=False)
    assert_close(va.numpy(), vr.numpy(), 0, 1e-12, 'verts')


def test_autograd_matches_finite_differences(body):
    m = ol.model_tensors(body, torch.float64)
    bp, go, be = [torch.tensor(a, dtype=torch.float64) for a in random_poses(1, 11)]
    w = torch.tensor(np.random.default_rng(0).standard_normal((1, body.num_verts, 3)))

    def f(bp_, be_):
        v, j = ol.smpl_forward(m, be_, bp_, go)
        return (v * w).sum() + j.sum()
    bp_ = bp.clone().requires_grad_(True)
    be_ = be.clone().requires_grad_(True)
    f(bp_, be_).backward()
    for idx in (0, 17, 44, 68):
        e = torch.zeros_like(bp)
        e[0, idx] = 1e-6
        fd = (f(bp + e, be) - f(bp - e, be)) / 2e-6
        assert_close(bp_.grad[0, idx].item(), fd.item(), 1e-5, 1e-7, 'd/dpose')
    for idx in (0, 9):
        e = torch.zeros_like(be)
        e[0, idx] = 1e-6
        fd = (f(bp, be + e) - f(bp, be - e)) / 2e-6
        assert_close(be_.grad[0, idx].item(), fd.item(), 1e-5, 1e-7, 'd/dbeta')
