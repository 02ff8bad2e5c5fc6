"""Caller-side geometry glue: estimate_translation and rotation_matrix_to_angle_axis.

CPU part: the oracle against the golden vectors produced by the reference's own functions
(tests/golden/make_golden_geometry.py) and against known answers.  GPU part (-m gpu): the HIP kernels
through the Python mirror against the same goldens and the oracle.
Tolerances: translations 1e-5 relative (float64 inside, float32 out); rotations: float32 round trips."""
import os

import numpy as np
import pytest
import torch

from oracle import geometry as og

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'geometry.npz'))


def rodrigues_np(aa):
    """Plain Rodrigues formula in float64 (independent of the code under test)."""
    out = []
    for v in np.asarray(aa, np.float64):
        th = np.linalg.norm(v)
        if th < 1e-12:
            out.append(np.eye(3))
            continue
        k = v / th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        out.append(np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K)
    return np.stack(out)


def branch_cases():
    """Angle-axis vectors that exercise all four branches of the quaternion extraction."""
    rng = np.random.default_rng(5)
    aa = [np.zeros(3), [np.pi - 1e-3, 0, 0], [0, np.pi - 1e-3, 0], [0, 0, np.pi - 1e-3], [3.0, 0.2, 0.1], [0.2, 3.0, 0.1],
          [0.1, 0.2, 3.0], [1e-4, 0, 0], [0.7, -0.7, 0.2]]
    aa += list(rng.standard_normal((200, 3)) * 1.3)
    aa = np.asarray(aa, np.float64)
    n = np.linalg.norm(aa, axis=1)
    aa[n > np.pi - 1e-3] *= ((np.pi - 1e-3) / n[n > np.pi - 1e-3])[:, None]     # keep the principal branch
    return aa


def test_oracle_estimate_translation_matches_reference():
    for key, f, img in (('trans', 5000.0, 224.0), ('trans_f1000', 1000.0, 256.0)):
        got = og.estimate_translation(G['S'], G['kp'], f, img, G['anno'])
        np.testing.assert_allclose(got, G[key], rtol=1e-6, atol=1e-7)
    assert (G['trans'][3] == 0).all() and (G['trans'][5] == 0).all()       # samples without confident joints


def test_oracle_rotmat_to_angle_axis_known_answers():
    aa = branch_cases()
    r = rodrigues_np(aa).astype(np.float32)
    got = og.rotation_matrix_to_angle_axis(r)
    np.testing.assert_allclose(got, aa, rtol=0, atol=3e-4)       # near pi the float32 matrix loses ~1e-4 of angle
    small = np.linalg.norm(aa, axis=1) < 2.5
    np.testing.assert_allclose(got[small], aa[small], rtol=0, atol=2e-6)
    hom = np.concatenate([r, np.tile(np.array([0, 0, 1], np.float32).reshape(1, 3, 1), (len(r), 1, 1))], 2)
    assert np.array_equal(og.rotation_matrix_to_angle_axis(hom), got)      # the callers' 3x4 form
    # all four branches were taken
    t = np.transpose(r, (0, 2, 1))
    d2, a, b = t[:, 2, 2] < 1e-6, t[:, 0, 0] > t[:, 1, 1], t[:, 0, 0] < -t[:, 1, 1]
    assert (d2 & a).any() and (d2 & ~a).any() and (~d2 & b).any() and (~d2 & ~b).any()
    # the selected trace term is always >= 1, so only NaN inputs give NaN (zeroed by the callers, train_module.py:212)
    bad = np.full((1, 3, 3), np.nan, np.float32)
    assert np.isnan(og.rotation_matrix_to_angle_axis(bad)).all()
    assert np.isfinite(og.rotation_matrix_to_angle_axis(-2 * np.eye(3, dtype=np.float32)[None])).all()


def test_mirror_rot6d_and_rodrigues_match_reference():
    from tuch_amd.utils import geometry as mg
    np.testing.assert_allclose(mg.rot6d_to_rotmat(torch.tensor(G['x6'])).numpy(), G['rot6d'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(mg.batch_rodrigues(torch.tensor(G['aa'])).numpy(), G['rodrigues'], rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_gpu_estimate_translation_matches_reference():
    from tuch_amd.utils import geometry as mg
    dev = torch.device('cuda:0')
    for key, f, img in (('trans', 5000.0, 224.0), ('trans_f1000', 1000.0, 256.0)):
        got = mg.estimate_translation(torch.tensor(G['S'], device=dev), torch.tensor(G['kp'], device=dev),
                                      focal_length=f, img_size=img, has_2d_kp_anno=torch.tensor(G['anno'], device=dev))
        assert got.device.type == 'cuda' and got.dtype == torch.float32
        np.testing.assert_allclose(got.cpu().numpy(), G[key], rtol=1e-5, atol=1e-6)
    # a large batch against the oracle
    rng = np.random.default_rng(2)
    b = 777
    S = (rng.standard_normal((b, 49, 3)) * 0.4).astype(np.float32)
    kp = np.concatenate([rng.uniform(0, 224, (b, 49, 2)), rng.uniform(0, 1, (b, 49, 1))], 2).astype(np.float32)
    anno = rng.uniform(size=b) < 0.5
    got = mg.estimate_translation(torch.tensor(S, device=dev), torch.tensor(kp, device=dev), 5000.0, 224.0,
                                  torch.tensor(anno, device=dev)).cpu().numpy()
    want = og.estimate_translation(S, kp, 5000.0, 224.0, anno)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_gpu_rotmat_to_angle_axis_matches_oracle():
    from tuch_amd.utils import geometry as mg
    dev = torch.device('cuda:0')
    aa = branch_cases()
    r = rodrigues_np(aa).astype(np.float32)
    want = og.rotation_matrix_to_angle_axis(r)
    got = mg.rotation_matrix_to_angle_axis(torch.tensor(r, device=dev)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
    near_pi = np.linalg.norm(aa, axis=1) > 3.0
    np.testing.assert_allclose(got[~near_pi], aa[~near_pi], rtol=0, atol=5e-5)
    hom = torch.cat([torch.tensor(r, device=dev), torch.tensor([0, 0, 1.0], device=dev).view(1, 3, 1).expand(len(r), -1, -1)], -1)
    assert torch.equal(mg.rotation_matrix_to_angle_axis(hom).cpu(), torch.tensor(got))
    # the way the reference uses it (train_module.py:207-212): rotmats from the regressor -> pose vector
    pose = (np.random.default_rng(9).standard_normal((8 * 24, 3)) * 0.4).astype(np.float32)
    rm = mg.batch_rodrigues(torch.tensor(pose, device=dev))
    back = mg.rotation_matrix_to_angle_axis(rm).view(8, -1)
    back[torch.isnan(back)] = 0.0
    np.testing.assert_allclose(back.cpu().numpy().reshape(-1, 3), pose, rtol=0, atol=5e-6)
    bad = torch.full((1, 3, 3), float('nan'), device=dev)
    assert torch.isnan(mg.rotation_matrix_to_angle_axis(bad)).all()


def test_angle_axis_to_rotation_matrix_known_answers():
    from tuch_amd.utils import geometry as mg
    aa = branch_cases()
    got = mg.angle_axis_to_rotation_matrix(torch.tensor(aa, dtype=torch.float32)).numpy()
    assert got.shape == (len(aa), 4, 4)
    big = np.linalg.norm(aa, axis=1) > 1e-2
    np.testing.assert_allclose(got[big, :3, :3], rodrigues_np(aa)[big], rtol=0, atol=5e-6)
    np.testing.assert_allclose(got[~big, :3, :3], rodrigues_np(aa)[~big], rtol=0, atol=1e-6)
    assert (got[:, 3, :3] == 0).all() and (got[:, :3, 3] == 0).all() and (got[:, 3, 3] == 1).all()
