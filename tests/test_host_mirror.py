"""CPU: the host-side mirror of the reference interface (geometry, prior, SPIN losses,
segment tables) against golden vectors from the reference."""
import numpy as np
import pytest
import torch

import golden_io as gio
from helpers import assert_close, golden


@pytest.fixture(scope='module')
def g():
    return golden('medium')


def _prior(g):
    from tuch_amd.smplify.prior import MaxMixturePrior
    return MaxMixturePrior(num_gaussians=8, gmm={k: g['gmm_' + k] for k in ('means', 'covars', 'weights')})


def test_projection_gmof_prior(g):
    from tuch_amd.smplify.losses import gmof
    from tuch_amd.utils.geometry import perspective_projection
    b = g['verts'].shape[0]
    proj = perspective_projection(torch.tensor(g['model_joints']), torch.eye(3)[None].expand(b, -1, -1),
                                  torch.tensor(g['camera_t']), 5000., torch.tensor(g['camera_center']))
    assert_close(proj.numpy(), g['projected_joints'], 1e-5, 1e-3, 'projection')
    gm = gmof(torch.tensor(g['joints_2d']) - torch.tensor(g['projected_joints']), 100.)
    assert_close(gm.numpy(), g['gmof_values'], 1e-5, 1e-5, 'gmof')
    pr = _prior(g)(torch.tensor(g['body_pose']), torch.tensor(g['betas']))
    assert_close(pr.numpy(), g['prior_values'], 1e-4, 1e-4, 'prior')


def test_camera_and_body_fitting_losses(g):
    import types
    from tuch_amd.smplify.losses import body_fitting_loss, camera_fitting_loss
    t = torch.tensor
    out = types.SimpleNamespace(joints=t(g['model_joints']), betas=t(g['betas']))
    cam = camera_fitting_loss(out, t(g['camera_t']), t(g['camera_t_est']), t(g['camera_center']),
                              t(g['joints_2d']), t(g['joints_conf']), focal_length=5000., shape_prior_weight=1.0)
    assert_close(cam.item(), g['camera_fitting_loss'], 1e-4, 0, 'camera_fitting_loss')
    pr = _prior(g)
    rep = body_fitting_loss(t(g['body_pose']), t(g['betas']), t(g['model_joints']), t(g['camera_t']),
                            t(g['camera_center']), t(g['joints_2d']), t(g['joints_conf']), pr,
                            focal_length=5000., output='reprojection')
    assert_close(rep.numpy(), g['body_fitting_reprojection'], 1e-4, 1e-4, 'reprojection')
    tot = body_fitting_loss(t(g['body_pose']), t(g['betas']), t(g['model_joints']), t(g['camera_t']),
                            t(g['camera_center']), t(g['joints_2d']), t(g['joints_conf']), pr, focal_length=5000.)
    assert_close(tot.item(), g['body_fitting_sum'], 1e-4, 0, 'body_fitting_loss')


def test_batch_rodrigues_is_a_rotation():
    from tuch_amd.utils.geometry import batch_rodrigues
    aa = torch.tensor(np.random.default_rng(0).standard_normal((16, 3)), dtype=torch.float32)
    r = batch_rodrigues(aa)
    assert_close((r @ r.transpose(1, 2)).numpy(), np.tile(np.eye(3), (16, 1, 1)), 0, 1e-5, 'orthonormal')
    assert_close(torch.det(r).numpy(), np.ones(16), 0, 1e-5, 'det')
    ang = torch.acos(((r.diagonal(dim1=1, dim2=2).sum(1) - 1) / 2).clamp(-1, 1))
    assert_close(ang.numpy(), aa.norm(dim=1).numpy(), 1e-4, 1e-4, 'angle')


@pytest.mark.parametrize('tag', ['small', 'medium'])
def test_segment_face_tables_match_reference(tag):
    from tuch_amd.utils.segmentation import BatchBodySegment
    gg = golden(tag)
    segs = gio.unpack_segments(gg)
    bbs = BatchBodySegment(list(segs.keys()), torch.tensor(gg['faces']), segs)
    want = gio.unpack_ragged('segment_faces', gg)
    for name, w in zip(segs.keys(), want):
        assert np.array_equal(bbs.segmentation[name].segment_faces.numpy().ravel(), w)
        assert bbs.segmentation[name].append_idx == gg['faces'].max()


def test_bench_reads_its_constants_from_the_committed_profiles():
    """bench.py's roofline.traffic / valu_busy and graph_timeline come from the newest summaries under profiles/ (not
    literals): the files parse, name the kernels the step runs today, and are internally consistent."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    bench = importlib.import_module('bench')
    prof = bench.profile_constants()
    for key in ('search', 'ray_leaf_kernel'):
        c = prof[key]
        assert c['kernel'] and c['source'].startswith('profiles/r'), c
        assert 1e6 < c['traffic_bytes'] < 1e9 and 0.2 < c['valu_busy'] <= 1.0 and c['valu_instr'] > 1e6, c
    assert prof['search']['kernel'].startswith('v2v_')
    tl = bench.timeline_constants()
    for key in ('batch64', 'batch8'):
        t = tl[key]
        assert t['kernels'] >= 10 and t['source'].startswith('profiles/r')
        assert abs(t['head_us'] + t['middle_us'] + t['tail_us'] - t['wall_us']) < 0.5
        assert 0 < t['wall_us'] < t['summed_kernel_us'] * 1.2
    assert tl['batch8']['wall_us'] < tl['batch64']['wall_us']
