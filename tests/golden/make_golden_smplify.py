#!/usr/bin/env python3
"""Golden vectors for the SMPLify-DC loop produced by running the
REFERENCE'S OWN ``SMPLifyDC.__call__`` (tuch/smplify/smplifydc.py:68-236, imported from
/root/reference, never copied) on a small synthetic body, CPU, 10 + 10 Adam iterations.

Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_smplify.py

The reference's loop cannot be imported as is: it needs the third-party
``smplx`` package, the un-shipped ``data.essentials.constants`` and a GPU.  What is stubbed:

  * ``smplx`` -- a module whose ``SMPL`` class is the body model of oracle/lbs.py (the restatement
    of smplx 0.1.13's published lbs / VertexJointSelector; PARITY UNPINNED at that boundary, as
    everywhere in this repo) and ``smplx.lbs.vertices2joints``.  The reference's own subclass
    ``tuch.models.smpl.SMPL`` (extra joint regressor from config.JOINT_REGRESSOR_TRAIN_EXTRA,
    joint map from constants) runs unmodified on top of it.
  * ``data.essentials.constants`` -- SPIN's joint tables (names, ids) with JOINT_MAP pointing at the
    synthetic body's 49-entry joint map.
  * ``trimesh`` / ``segm_utils`` -- as in make_golden.py (contents = our synthetic segments).
  * F7 shims: ``torch.cuda.LongTensor``; ``contact_fitting_loss`` is given ``device='cpu'``
    (losses.py:43 defaults to 'cuda'; smplifydc.py:162 does not pass it).

Everything else -- the two Adam loops, which parameters are optimised in which stage, the
confidence zeroing, camera_fitting_loss / contact_fitting_loss / body_fitting_loss, the GMM prior,
the final evaluation and the returned 7-tuple -- is the reference's code.
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

import golden_io as gio
from oracle import lbs as olbs
from tuch_amd.models.smpl import SPIN_JOINT_NAMES
from synthetic import make_body, random_poses

torch.cuda.LongTensor = torch.LongTensor
_STATE = {'body': None}


# ------------------------------------------------------------------------------ stubs
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
    const.FOCAL_LENGTH = 5000.
    const.IMG_RES = 224
    const.JOINT_NAMES = list(SPIN_JOINT_NAMES)
    const.JOINT_IDS = {n: i for i, n in enumerate(const.JOINT_NAMES)}
    const.JOINT_MAP = {}
    sys.modules[const.__name__] = const
    sys.modules['data.essentials'].constants = const

    # smplx: SMPL on the oracle LBS
    smplx = types.ModuleType('smplx')
    smplx.__path__ = []
    lbs_mod = types.ModuleType('smplx.lbs')
    lbs_mod.vertices2joints = lambda J_regressor, vertices: torch.einsum('bik,ji->bjk', [vertices, J_regressor])
    Out = namedtuple('SMPLOutput', ['vertices', 'joints', 'full_pose', 'betas', 'global_orient', 'body_pose'])

    class SMPL(nn.Module):
        def __init__(self, model_path, batch_size=1, create_transl=True, **kwargs):
            super().__init__()
            body = _STATE['body']
            self.faces = body.faces
            self.m = olbs.model_tensors(body)

        def get_num_verts(self):
            return self.m['v_template'].shape[0]

        def forward(self, betas=None, body_pose=None, global_orient=None, get_skin=True,
                    return_full_pose=False, pose2rot=True, **kwargs):
            full = torch.cat([global_orient.reshape(betas.shape[0], -1), body_pose.reshape(betas.shape[0], -1)], 1)
            verts, joints = olbs.lbs(betas, full, self.m, pose2rot)
            joints = torch.cat([joints, verts[:, self.m['extra_vertex_ids']]], 1)     # VertexJointSelector
            return Out(vertices=verts, joints=joints, full_pose=full if return_full_pose else None,
                       betas=betas, global_orient=global_orient, body_pose=body_pose)
    smplx.SMPL = SMPL
    smplx.lbs = lbs_mod
    sys.modules['smplx'] = smplx
    sys.modules['smplx.lbs'] = lbs_mod
    return su, const


_SEGM_UTILS, _CONST = _install_stubs()

from configs import config as ref_config                      # noqa: E402
from tuch.smplify import losses as ref_losses                 # noqa: E402
from tuch.smplify import smplifydc as ref_smplifydc           # noqa: E402
from tuch.utils import geometry as ref_geometry               # noqa: E402
from tuch.utils import segmentation as ref_segmentation       # noqa: E402

# F7: smplifydc.py:162 calls contact_fitting_loss without device= (default 'cuda')
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
    rings, segs, batch, iters, seed = 14, 16, 3, 10, 4004
    body = make_body(rings, segs, relax_iters=40)
    _use_body(body)
    out = {'rings': np.int64(rings), 'segs': np.int64(segs), 'relax_iters': np.int64(40),
           'num_iters': np.int64(iters), 'geothres': np.float32(0.3), 'euclthres': np.float32(0.02),
           'contact_loss_weight': np.float32(2000.0)}
    bp, go, be = random_poses(batch, seed)
    rng = np.random.Generator(np.random.PCG64(seed))
    num_pairs = len(body.region_pairs)
    gt = (rng.random((batch, num_pairs)) < 0.05).astype(np.float32)
    gt[:, 1] = 1.0
    has_dc = np.array([True, False, True])
    ignore = np.array([False, False, True])
    has_gt_kp = np.array([False, True, False])
    cam_t = (np.tile([[0., 0., 20.]], (batch, 1)) + 0.2 * rng.standard_normal((batch, 3))).astype(np.float32)
    cam_c = np.zeros((batch, 2), np.float32)
    with tempfile.TemporaryDirectory() as tmp:
        ref_config.PRIOR_FOLDER = tmp
        ref_config.JOINT_REGRESSOR_TRAIN_EXTRA = os.path.join(tmp, 'J_regressor_extra.npy')
        np.save(ref_config.JOINT_REGRESSOR_TRAIN_EXTRA, body.J_regressor_extra)
        with open(os.path.join(tmp, 'gmm_08.pkl'), 'wb') as f:
            pickle.dump({k: np.asarray(v, np.float64) for k, v in body.gmm.items()}, f)
        fitter = ref_smplifydc.SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=iters, focal_length=5000.,
                                         geodistssmpl=torch.tensor(body.geodesics), geothres=0.3, euclthres=0.02,
                                         device=torch.device('cpu'))
    assert fitter.ign_joints == [1, 9, 12, 27, 28], fitter.ign_joints
    # keypoints = projection of a perturbed pose + noise, so that both stages have something to do
    with torch.no_grad():
        tgt = fitter.smpl(global_orient=torch.tensor(go), body_pose=torch.tensor(bp) + 0.1, betas=torch.tensor(be) * 0.5)
        j2d = ref_geometry.perspective_projection(tgt.joints, torch.eye(3)[None].expand(batch, -1, -1),
                                                  torch.tensor(cam_t) + 0.3, 5000., torch.tensor(cam_c)).numpy()
    j2d = j2d + 2.0 * rng.standard_normal(j2d.shape).astype(np.float32)
    conf = (0.5 + 0.5 * rng.random((batch, 49, 1))).astype(np.float32)
    kp = np.concatenate([j2d.astype(np.float32), conf], 2)
    init_pose = np.concatenate([go, bp], 1).astype(np.float32)
    out.update(init_pose=init_pose, init_betas=be, init_cam_t=cam_t, camera_center=cam_c, keypoints_2d=kp,
               gt_contact=gt, has_discrete_contact=has_dc, ignore_idxs=ignore, has_gt_keypoints=has_gt_kp)
    face_tensor = torch.tensor(body.faces, dtype=torch.long)
    segments = ref_segmentation.BatchBodySegment(list(body.segments.keys()), face_tensor)
    cdict = {'classes': [list(p) for p in body.region_pairs], 'csig': dict(body.regions)}
    names = ('vertices', 'joints', 'pose', 'betas', 'camera_translation', 'reprojection_loss')
    for tag, use_contact in (('contact', True), ('plain', False)):
        res = fitter(torch.tensor(init_pose), torch.tensor(be), torch.tensor(cam_t), torch.tensor(cam_c),
                     torch.tensor(kp), use_contact=use_contact, contactlist=cdict,
                     gt_contact=[torch.tensor(gt), None], ignore_idxs=torch.tensor(ignore),
                     has_discrete_contact=torch.tensor(has_dc), has_gt_keypoints=torch.tensor(has_gt_kp),
                     contact_loss_weight=2000.0, segments=segments)
        for n, t in zip(names, res[:6]):
            out['%s_%s' % (tag, n)] = t.detach().numpy()
        out['%s_optiverts' % tag] = torch.stack([v.detach() for v in res[6]]).numpy()
        print(tag, 'final reprojection', float(res[5].sum()), 'moved',
              float((res[2] - torch.tensor(init_pose)).abs().max()))
    # get_fitting_loss (smplifydc.py:238-276) on the initial parameters
    out['get_fitting_loss'] = fitter.get_fitting_loss(torch.tensor(init_pose), torch.tensor(be), torch.tensor(cam_t),
                                                      torch.tensor(cam_c), torch.tensor(kp),
                                                      torch.tensor(has_gt_kp)).numpy()
    path = os.path.join(HERE, 'smplify_loop.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
