"""RegressorLoss.forward (tuch/train/loss.py:94-168): the seven loss-dict entries and the total against golden
vectors produced by the reference's own class (tests/golden/make_golden_regressor.py).  contact_loss_weight = 0
in the fixture (the contact term has its own goldens), so this runs on the CPU: the SPIN terms are torch ops."""
import os
import types

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'regressor_forward.npz'))
ORDER = ['pred_rotmat', 'pred_betas', 'opt_pose', 'opt_betas', 'pred_keypoints_2d', 'gt_keypoints_2d', 'pred_joints',
         'gt_joints', 'has_pose_3d', 'pred_vertices', 'opt_vertices', 'pred_camera', 'valid_fit', 'valid_fit_shape']
KEYS = ['loss_shape', 'loss_keypoints', 'loss_keypoints_3d', 'loss_regr_pose', 'loss_regr_betas', 'loss_cam', 'loss_contact']


def make_criterion():
    from tuch_amd.train.loss import RegressorLoss
    o = G['options']
    crit = RegressorLoss.__new__(RegressorLoss)            # the constructor builds the device model; not needed here
    torch.nn.Module.__init__(crit)
    crit.device = torch.device('cpu')
    crit.options = types.SimpleNamespace(contact_loss_weight=0.0, shape_loss_weight=o[0], keypoint_loss_weight=o[1],
                                         pose_loss_weight=o[2], beta_loss_weight=o[3], openpose_train_weight=o[4],
                                         gt_train_weight=o[5])
    crit.criterion_shape = torch.nn.L1Loss()
    crit.criterion_keypoints = torch.nn.MSELoss(reduction='none')
    crit.criterion_regr = torch.nn.MSELoss()
    return crit


@pytest.mark.parametrize('tag', ['a', 'none_valid'])
def test_forward_matches_reference(tag):
    crit = make_criterion()
    args = []
    for k in ORDER:
        key = '%s_in_%s' % (tag, k)
        args.append(torch.tensor(G[key] if key in G.files else G['in_' + k]))
    total, d = crit.forward(*args)
    assert list(d.keys()) == KEYS
    for k in KEYS:
        got = float(torch.as_tensor(d[k], dtype=torch.float64).reshape(-1)[0])
        np.testing.assert_allclose(got, float(G['%s_%s' % (tag, k)]), rtol=2e-6, atol=1e-9, equal_nan=True, err_msg=k)
    np.testing.assert_allclose(float(total.reshape(-1)[0]), float(G[tag + '_total']), rtol=2e-6, equal_nan=True)


def test_forward_backward_reaches_every_prediction():
    crit = make_criterion()
    args = [torch.tensor(G['in_' + k]) for k in ORDER]
    grads = {}
    for i, k in enumerate(ORDER):
        if k.startswith('pred_'):
            args[i] = args[i].clone().requires_grad_(True)
            grads[k] = args[i]
    total, _ = crit.forward(*args)
    total.backward()
    for k, t in grads.items():
        assert t.grad is not None and torch.isfinite(t.grad).all() and t.grad.abs().sum() > 0, k


def test_eft_forward_matches_reference():
    """EFTLoss.forward (tuch/eft/loss.py:73-117) with contact_weight = 0: keypoint and shape terms, x60 total."""
    from tuch_amd.eft.loss import EFTLoss
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'eft_forward.npz'))
    crit = EFTLoss.__new__(EFTLoss)
    torch.nn.Module.__init__(crit)
    crit.device, crit.options = torch.device('cpu'), types.SimpleNamespace(img_res=224)
    crit.focal_length, crit.camera_center = 5000, torch.zeros(2)
    crit.criterion_keypoints = torch.nn.MSELoss(reduction='none')
    crit.keypoints_weight, crit.shape_weight, crit.contact_weight = [float(x) for x in g['weights']]
    body = types.SimpleNamespace(joints=torch.tensor(g['joints']), betas=torch.tensor(g['betas']), vertices=None)
    loss, d = crit.forward(body, torch.tensor(g['camera']), {'keypoints': torch.tensor(g['keypoints']), 'contact': None})
    assert list(d.keys()) == ['loss_shape', 'loss_keypoints', 'loss_contact']
    for k in d:
        np.testing.assert_allclose(float(d[k]), float(g[k]), rtol=2e-6, atol=1e-9, err_msg=k)
    np.testing.assert_allclose(float(loss), float(g['total']), rtol=2e-6)
