#!/usr/bin/env python3
"""Golden vectors for the CALLERS of the hot path -- BASELINE configs 4/5 -- produced by running the REFERENCE'S OWN
``TUCH.forward_train_step`` (tuch/train/train_module.py:105-335) and ``FitsDict`` (tuch/train/fits_dict.py) on CPU:
the training step with SMPLify-DC in the loop (run_smplify, use_contact_in_the_loop), the dictionary of best fits,
the valid-fit logic and the reference's RegressorLoss (HD branch) -- all of it the reference's code.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_train.py

Stubs (the packages are absent here and not vendored):
  * ``smplx`` (SMPL on oracle/lbs.py), ``data.essentials.constants``, ``trimesh`` / ``segm_utils``: as in
    make_golden_smplify.py;
  * ``torchgeometry``: rotation_matrix_to_angle_axis = oracle/geometry.py (restated from the published 0.1.2
    algorithm, PARITY UNPINNED), angle_axis_to_rotation_matrix = the restatement in tuch_amd/utils/geometry.py (pure
    torch, runs on CPU);
  * ``cv2``: only ``cv2.Rodrigues(R)`` is used (fits_dict.py:115-117): stood in by scipy's
    ``Rotation.from_matrix(R).as_rotvec()``, an independent implementation of the same function;
  * the two regressors (HMR, SPIN) are synthetic.make_regressor stand-ins, the datasets object only carries
    ``dataset_dict`` / ``datasets`` (all that FitsDict reads);
  * F7 shims: torch.cuda.LongTensor, contact_fitting_loss(device='cpu'), SMPLifyDC(device='cpu').
"""
import functools
import os
import pickle
import sys
import tempfile
import types
from collections import namedtuple

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(1, REF)

import numpy as np
import torch
import torch.nn as nn
from scipy.spatial.transform import Rotation

from oracle import geometry as ogeo
from oracle import lbs as olbs
from tuch_amd.models.smpl import SPIN_JOINT_NAMES
from synthetic import dense_hd_regressor, make_body, make_regressor, make_train_batch
from tuch_amd.train.fits_dict import SMPL_JOINTS_FLIP_PERM, SMPL_POSE_FLIP_PERM
from tuch_amd.utils import geometry as our_geometry

torch.cuda.LongTensor = torch.LongTensor
_STATE = {'body': None}


def _install_stubs():
    tm = types.ModuleType('trimesh')

    def load(path, process=False):
        body = _STATE['body']
        name = os.path.basename(path)[len('smpl_segment_'):-len('.ply')]
        colors = np.zeros((body.num_verts, 4), np.uint8)
        colors[body.segments[name]['vidx'], 0] = 255
        return types.SimpleNamespace(visual=types.SimpleNamespace(vertex_colors=colors))
    tm.load = load
    sys.modules['trimesh'] = tm
    for name in ['data', 'data.essentials', 'data.essentials.segments', 'data.essentials.segments.smpl']:
        mod = types.ModuleType(name)
        mod.__path__ = []
        sys.modules[name] = mod
    su = types.ModuleType('data.essentials.segments.smpl.segm_utils')
    su.segments = {}
    sys.modules[su.__name__] = su
    sys.modules['data.essentials.segments.smpl'].segm_utils = su
    const = types.ModuleType('data.essentials.constants')
    const.FOCAL_LENGTH, const.IMG_RES = 5000., 224
    const.JOINT_NAMES = list(SPIN_JOINT_NAMES)
    const.JOINT_IDS = {n: i for i, n in enumerate(const.JOINT_NAMES)}
    const.JOINT_MAP = {}
    const.SMPL_JOINTS_FLIP_PERM = list(SMPL_JOINTS_FLIP_PERM)
    const.SMPL_POSE_FLIP_PERM = list(SMPL_POSE_FLIP_PERM)
    sys.modules[const.__name__] = const
    sys.modules['data.essentials'].constants = const
    smplx = types.ModuleType('smplx')
    smplx.__path__ = []
    lbs_mod = types.ModuleType('smplx.lbs')
    lbs_mod.vertices2joints = lambda J_regressor, vertices: torch.einsum('bik,ji->bjk', [vertices, J_regressor])
    Out = namedtuple('SMPLOutput', ['vertices', 'joints', 'full_pose', 'betas', 'global_orient', 'body_pose'])

    class SMPL(nn.Module):
        def __init__(self, model_path, batch_size=1, create_transl=True, **kwargs):
            super().__init__()
            self.faces = _STATE['body'].faces
            self.m = olbs.model_tensors(_STATE['body'])

        def get_num_verts(self):
            return self.m['v_template'].shape[0]

        def forward(self, betas=None, body_pose=None, global_orient=None, get_skin=True, return_full_pose=False,
                    pose2rot=True, **kwargs):
            n = betas.shape[0]
            if pose2rot:
                full = torch.cat([global_orient.reshape(n, -1), body_pose.reshape(n, -1)], 1)
            else:
                full = torch.cat([global_orient.reshape(n, 1, 3, 3), body_pose.reshape(n, 23, 3, 3)], 1)
            verts, joints = olbs.lbs(betas, full, self.m, pose2rot)
            joints = torch.cat([joints, verts[:, self.m['extra_vertex_ids']]], 1)
            return Out(vertices=verts, joints=joints, full_pose=full if return_full_pose else None, betas=betas,
                       global_orient=global_orient, body_pose=body_pose)
    smplx.SMPL, smplx.lbs = SMPL, lbs_mod
    sys.modules['smplx'], sys.modules['smplx.lbs'] = smplx, lbs_mod
    tg = types.ModuleType('torchgeometry')
    tg.rotation_matrix_to_angle_axis = lambda r: torch.tensor(ogeo.rotation_matrix_to_angle_axis(r.detach().numpy()))
    tg.angle_axis_to_rotation_matrix = our_geometry.angle_axis_to_rotation_matrix
    sys.modules['torchgeometry'] = tg
    cv2 = types.ModuleType('cv2')
    cv2.Rodrigues = lambda R: (Rotation.from_matrix(np.asarray(R, np.float64)).as_rotvec().reshape(3, 1), None)
    sys.modules['cv2'] = cv2
    return su, const


_SEGM_UTILS, _CONST = _install_stubs()

from configs import config as ref_config                      # noqa: E402
from tuch.smplify import losses as ref_losses                 # noqa: E402
from tuch.smplify import smplifydc as ref_smplifydc           # noqa: E402
from tuch.train import loss as ref_train_loss                 # noqa: E402
from tuch.train import train_module as ref_train_module       # noqa: E402
from tuch.train.fits_dict import FitsDict as RefFitsDict      # noqa: E402

ref_smplifydc.contact_fitting_loss = functools.partial(ref_losses.contact_fitting_loss, device='cpu')


def _use_body(body):
    _STATE['body'] = body
    _SEGM_UTILS.segments.clear()
    for name, seg in body.segments.items():
        _SEGM_UTILS.segments[name] = {k: [int(x) for x in v] for k, v in seg['bands'].items()}
    _CONST.JOINT_MAP.clear()
    for i, n in enumerate(_CONST.JOINT_NAMES):
        _CONST.JOINT_MAP[n] = int(body.joint_map[i])


def main():
    rings, segs, batch, seed = 14, 16, 6, 6006
    body = make_body(rings, segs, relax_iters=40)
    _use_body(body)
    datasets = (('dsA', 50), ('dsB', 30))
    raw = make_train_batch(body, batch, seed, datasets)
    out = {'rings': np.int64(rings), 'segs': np.int64(segs), 'relax_iters': np.int64(40), 'seed': np.int64(seed),
           'batch': np.int64(batch)}
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    static = {n: np.concatenate([0.2 * rng.standard_normal((k, 72)), 0.5 * rng.standard_normal((k, 10))], 1).astype(np.float32)
              for n, k in datasets}
    for n in static:
        out['static_fits_' + n] = static[n]
    options = types.SimpleNamespace(
        batch_size=batch, img_res=224, run_smplify=True, use_contact_in_the_loop=True, contact_in_the_loop_loss_weight=2000.0,
        smplify_threshold=100.0, num_smplify_iters=5, contact_loss_weight=1.0, shape_loss_weight=0.5, keypoint_loss_weight=5.0,
        pose_loss_weight=1.0, beta_loss_weight=0.001, openpose_train_weight=0.0, gt_train_weight=1.0, checkpoint_dir=None)
    with tempfile.TemporaryDirectory() as tmp:
        options.checkpoint_dir = tmp
        ref_config.PRIOR_FOLDER = tmp
        ref_config.STATIC_FITS_DIR = tmp
        ref_config.HD_MODEL_DIR = tmp
        ref_config.DSC_ROOT = tmp
        ref_config.JOINT_REGRESSOR_TRAIN_EXTRA = os.path.join(tmp, 'J_regressor_extra.npy')
        np.save(ref_config.JOINT_REGRESSOR_TRAIN_EXTRA, body.J_regressor_extra)
        with open(os.path.join(tmp, 'gmm_08.pkl'), 'wb') as f:
            pickle.dump({k: np.asarray(v, np.float64) for k, v in body.gmm.items()}, f)
        np.save(os.path.join(tmp, 'smpl_neutral_hd_vert_regressor.npy'), dense_hd_regressor(body))
        with open(os.path.join(tmp, 'smpl_neutral_hd_sample_from_mesh_out.pkl'), 'wb') as f:
            pickle.dump({'faces_vert_is_sampled_from': body.hd_face_id}, f)
        with open(os.path.join(tmp, 'classes.pkl'), 'wb') as f:
            pickle.dump(np.asarray(body.region_pairs), f)
        with open(os.path.join(tmp, 'ContactSigSMPL.pkl'), 'wb') as f:
            pickle.dump({k: [int(x) for x in v] for k, v in body.regions.items()}, f)
        for n in static:
            np.save(os.path.join(tmp, n + '_fits.npy'), static[n])
        train_ds = types.SimpleNamespace(dataset_dict={n: i for i, (n, _) in enumerate(datasets)},
                                         datasets=[list(range(k)) for _, k in datasets])
        from tuch.models.smpl import SMPL
        smpl = SMPL(ref_config.SMPL_MODEL_DIR, batch_size=batch, create_transl=False)
        face_tensor = torch.tensor(body.faces.astype(np.int64))[None].repeat(batch, 1, 1)
        geod = torch.tensor(body.geodesics)
        smplify = ref_smplifydc.SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=options.num_smplify_iters,
                                          focal_length=5000., geodistssmpl=geod, geothres=0.3, euclthres=0.02,
                                          device=torch.device('cpu'))
        criterion = ref_train_loss.RegressorLoss(options=options, device='cpu', num_verts=body.num_verts, faces=face_tensor,
                                                 geodistssmpl=geod, geothres=0.3, face_tensor=face_tensor)
        module = ref_train_module.TUCH(options=options, device='cpu', datasets=(train_ds, None), bodymodel=smpl,
                                       spin_model=make_regressor(11), regressor=make_regressor(12),
                                       optimization=smplify, criterion=criterion, geodistssmpl=geod)
        batch_t = {k: (torch.tensor(v) if not isinstance(v, list) else v) for k, v in raw.items()}
        # ---- FitsDict on its own: gather (flip + rotation) and scatter back
        fd = RefFitsDict(options, train_ds)
        pose, betas = fd[(raw['dataset_name'], batch_t['sample_index'], batch_t['rot_angle'], batch_t['is_flipped'])]
        out['fits_get_pose'], out['fits_get_betas'] = pose.numpy(), betas.numpy()
        upd = np.array([True, False, True, True, False, True])
        new_pose = (pose + 0.05 * torch.tensor(rng.standard_normal(pose.shape), dtype=torch.float32))
        fd[(raw['dataset_name'], batch_t['sample_index'], batch_t['rot_angle'], batch_t['is_flipped'], torch.tensor(upd))] = \
            (new_pose, betas + 0.1)
        out['fits_set_update'], out['fits_set_pose'] = upd, new_pose.numpy()
        for n in static:
            out['fits_after_set_' + n] = fd.fits_dict[n].numpy().copy()
        # ---- the training step
        loss, losses, output = module.forward_train_step(batch_t)
        loss.backward()
        out['loss'] = np.float64(loss.item())
        for k, v in losses.items():
            out['losses_' + k] = np.asarray(v.detach().numpy(), np.float64)
        for k in ('pred_vertices', 'opt_vertices', 'pred_cam_t', 'opt_cam_t', 'spin_vertices', 'spin_cam_t', 'gt_keypoints'):
            out['output_' + k] = output[k].numpy()
        out['output_valid_kpts_anno'] = output['valid_kpts_anno'].numpy()
        out['output_smplifyoptiverts_last'] = output['smplifyoptiverts'][-1].detach().numpy()
        for n in static:
            out['fits_after_step_' + n] = module.fits_dict.fits_dict[n].numpy().copy()
        out['grad_fc_weight'] = module.model.fc.weight.grad.numpy()
        print('loss', loss.item(), {k: float(v) for k, v in losses.items()})
        print('valid', output['valid_kpts_anno'].tolist(),
              'rows changed', {n: int((out['fits_after_step_' + n] != static[n]).any(1).sum()) for n in static})
    for k, v in raw.items():
        out['batch_' + k] = np.asarray(v)
    for k, v in vars(options).items():
        if k != 'checkpoint_dir':
            out['opt_' + k] = np.asarray(v)
    path = os.path.join(HERE, 'train_step.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
