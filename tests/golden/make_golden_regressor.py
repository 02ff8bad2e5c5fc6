#!/usr/bin/env python3
"""Golden vectors for RegressorLoss.forward (tuch/train/loss.py:94-168 with its SPIN terms :172-238), produced by
the REFERENCE's own class (imported from /root/reference, never copied) on synthetic inputs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_regressor.py

contact_loss_weight = 0 here: the contact term has its own goldens (contact_*.npz); this fixture pins the seven
entries of the loss dict and the total that the regressor's backward starts from.  Writes regressor_forward.npz.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(1, '/root/reference')

import numpy as np
import torch

for name in ('trimesh', 'data', 'data.essentials', 'data.essentials.segments', 'data.essentials.segments.smpl',
             'data.essentials.segments.smpl.segm_utils'):
    mod = types.ModuleType(name)
    mod.__path__ = []
    sys.modules[name] = mod
sys.modules['data.essentials.segments.smpl.segm_utils'].segments = {}

from tuch.train import loss as ref                      # noqa: E402

rng = np.random.default_rng(21)
B, V = 6, 40
opts = types.SimpleNamespace(contact_loss_weight=0.0, shape_loss_weight=0.5, keypoint_loss_weight=5.0,
                             pose_loss_weight=1.0, beta_loss_weight=0.001, openpose_train_weight=0.0,
                             gt_train_weight=1.0)
crit = ref.RegressorLoss.__new__(ref.RegressorLoss)
torch.nn.Module.__init__(crit)
crit.device, crit.options = 'cpu', opts
crit.criterion_shape = torch.nn.L1Loss()
crit.criterion_keypoints = torch.nn.MSELoss(reduction='none')
crit.criterion_regr = torch.nn.MSELoss()


def rotmats(n):
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)


inp = dict(
    pred_rotmat=rotmats(B * 24).reshape(B, 24, 3, 3).astype(np.float32),
    pred_betas=rng.standard_normal((B, 10)).astype(np.float32),
    opt_pose=(0.3 * rng.standard_normal((B, 72))).astype(np.float32),
    opt_betas=rng.standard_normal((B, 10)).astype(np.float32),
    pred_keypoints_2d=rng.uniform(-1, 1, (B, 49, 2)).astype(np.float32),
    gt_keypoints_2d=np.concatenate([rng.uniform(-1, 1, (B, 49, 2)), rng.uniform(0, 1, (B, 49, 1))], 2).astype(np.float32),
    pred_joints=rng.standard_normal((B, 49, 3)).astype(np.float32),
    gt_joints=np.concatenate([rng.standard_normal((B, 24, 3)), rng.uniform(0, 1, (B, 24, 1))], 2).astype(np.float32),
    has_pose_3d=np.array([1, 0, 1, 1, 0, 0], np.uint8),
    pred_vertices=rng.standard_normal((B, V, 3)).astype(np.float32),
    opt_vertices=rng.standard_normal((B, V, 3)).astype(np.float32),
    pred_camera=rng.uniform(0.5, 1.5, (B, 3)).astype(np.float32),
    valid_fit=np.array([1, 1, 0, 1, 0, 1], bool),
    valid_fit_shape=np.array([1, 0, 0, 1, 1, 1], bool),
)
order = ['pred_rotmat', 'pred_betas', 'opt_pose', 'opt_betas', 'pred_keypoints_2d', 'gt_keypoints_2d', 'pred_joints',
         'gt_joints', 'has_pose_3d', 'pred_vertices', 'opt_vertices', 'pred_camera', 'valid_fit', 'valid_fit_shape']
out = {}
for tag, mod in (('a', {}), ('none_valid', dict(valid_fit=np.zeros(B, bool), valid_fit_shape=np.zeros(B, bool),
                                                has_pose_3d=np.zeros(B, np.uint8)))):
    cur = dict(inp, **mod)
    args = [torch.tensor(cur[k]) for k in order]
    with np.errstate(all='ignore'):
        total, d = crit.forward(*args)
    out[tag + '_total'] = np.float64(total.reshape(-1)[0].item())
    for k, v in d.items():
        out[tag + '_' + k] = np.float64(torch.as_tensor(v, dtype=torch.float64).reshape(-1)[0].item())
    for k in ('valid_fit', 'valid_fit_shape', 'has_pose_3d'):
        out[tag + '_in_' + k] = cur[k]
out.update({'in_' + k: v for k, v in inp.items()})
out['options'] = np.array([opts.shape_loss_weight, opts.keypoint_loss_weight, opts.pose_loss_weight,
                           opts.beta_loss_weight, opts.openpose_train_weight, opts.gt_train_weight])
np.savez_compressed(os.path.join(HERE, 'regressor_forward.npz'), **out)
print({k: float(v) for k, v in out.items() if k.startswith('a_') and 'in_' not in k})
print({k: float(v) for k, v in out.items() if k.startswith('none_valid_') and 'in_' not in k})

# ---- EFTLoss.forward (tuch/eft/loss.py:73-117) with contact_weight = 0: keypoint + shape terms and the x60 total
import contextlib
import io
from tuch.eft import loss as ref_eft                    # noqa: E402

eft = ref_eft.EFTLoss.__new__(ref_eft.EFTLoss)
torch.nn.Module.__init__(eft)
eft.device, eft.options = 'cpu', types.SimpleNamespace(img_res=224)
eft.focal_length, eft.camera_center = 5000, torch.tensor([0, 0])
eft.criterion_keypoints = torch.nn.MSELoss(reduction='none')
eft.keypoints_weight, eft.shape_weight, eft.contact_weight = 0.7, 1.3, 0.0
Be = 3
joints = (0.4 * rng.standard_normal((Be, 49, 3))).astype(np.float32)
betas = rng.standard_normal((Be, 10)).astype(np.float32)
camera = np.stack([rng.uniform(0.6, 1.2, Be), rng.uniform(-0.1, 0.1, Be), rng.uniform(-0.1, 0.1, Be)], 1).astype(np.float32)
kp = np.concatenate([rng.uniform(-1, 1, (Be, 49, 2)), rng.uniform(0, 1, (Be, 49, 1))], 2).astype(np.float32)
body = types.SimpleNamespace(joints=torch.tensor(joints), betas=torch.tensor(betas), vertices=None)
with contextlib.redirect_stdout(io.StringIO()):          # the reference prints the losses (:115)
    loss, d = eft.forward(body, torch.tensor(camera), {'keypoints': torch.tensor(kp), 'contact': None})
eft_out = dict(joints=joints, betas=betas, camera=camera, keypoints=kp, weights=np.array([0.7, 1.3, 0.0]),
               total=np.float64(loss.item()), **{k: np.float64(v.item()) for k, v in d.items()})
np.savez_compressed(os.path.join(HERE, 'eft_forward.npz'), **eft_out)
print({k: float(v) for k, v in eft_out.items() if np.ndim(v) == 0})
