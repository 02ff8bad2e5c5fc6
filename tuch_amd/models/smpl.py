"""Drop-in for the reference's ``tuch/models/smpl.py``: SMPL linear blend skinning with 21
picked vertices + 9 extra regressed joints, re-mapped to the 49 joints the losses use.

The reference subclasses ``smplx.SMPL`` (third-party, absent here) and loads the
licensed model .pkl.  This class takes the reference's constructor call verbatim,

    SMPL(config.SMPL_MODEL_DIR, batch_size=..., create_transl=False)        # train.py:57

and then does what the reference + smplx do between them: SMPL_{GENDER}.pkl from the model
directory, ``J_regressor_extra`` from ``config.JOINT_REGRESSOR_TRAIN_EXTRA`` and the joint map
from ``constants.JOINT_MAP / JOINT_NAMES`` (tuch/models/smpl.py:37-42), the 21 picked vertices
from smplx's vertex-id table.  ``model_data=`` takes already loaded arrays instead (e.g.
synthetic.SyntheticBody).  Returns the same ``ModelOutput``.
"""
from __future__ import annotations

import importlib
import io
import os
import pickle
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

# SPIN's joint tables (data/essentials/constants.py is the un-shipped copy of SPIN's constants.py).
# Used only when that module cannot be imported; restated from the published SPIN repository and
# cross-checked against the reference's own uses: the ignored joints of smplifydc.py:46-47 come out
# as [1, 9, 12, 27, 28] and loss.py:197-200 takes the pelvis as the mean of ground-truth joints 2, 3
# (Right/Left Hip).
SPIN_JOINT_NAMES = [
    'OP Nose', 'OP Neck', 'OP RShoulder', 'OP RElbow', 'OP RWrist', 'OP LShoulder', 'OP LElbow',
    'OP LWrist', 'OP MidHip', 'OP RHip', 'OP RKnee', 'OP RAnkle', 'OP LHip', 'OP LKnee', 'OP LAnkle',
    'OP REye', 'OP LEye', 'OP REar', 'OP LEar', 'OP LBigToe', 'OP LSmallToe', 'OP LHeel', 'OP RBigToe',
    'OP RSmallToe', 'OP RHeel',
    'Right Ankle', 'Right Knee', 'Right Hip', 'Left Hip', 'Left Knee', 'Left Ankle', 'Right Wrist',
    'Right Elbow', 'Right Shoulder', 'Left Shoulder', 'Left Elbow', 'Left Wrist', 'Neck (LSP)',
    'Top of Head (LSP)', 'Pelvis (MPII)', 'Thorax (MPII)', 'Spine (H36M)', 'Jaw (H36M)', 'Head (H36M)',
    'Nose', 'Left Eye', 'Right Eye', 'Left Ear', 'Right Ear']
SPIN_JOINT_MAP = {
    'OP Nose': 24, 'OP Neck': 12, 'OP RShoulder': 17, 'OP RElbow': 19, 'OP RWrist': 21, 'OP LShoulder': 16,
    'OP LElbow': 18, 'OP LWrist': 20, 'OP MidHip': 0, 'OP RHip': 2, 'OP RKnee': 5, 'OP RAnkle': 8,
    'OP LHip': 1, 'OP LKnee': 4, 'OP LAnkle': 7, 'OP REye': 25, 'OP LEye': 26, 'OP REar': 27, 'OP LEar': 28,
    'OP LBigToe': 29, 'OP LSmallToe': 30, 'OP LHeel': 31, 'OP RBigToe': 32, 'OP RSmallToe': 33, 'OP RHeel': 34,
    'Right Ankle': 8, 'Right Knee': 5, 'Right Hip': 45, 'Left Hip': 46, 'Left Knee': 4, 'Left Ankle': 7,
    'Right Wrist': 21, 'Right Elbow': 19, 'Right Shoulder': 17, 'Left Shoulder': 16, 'Left Elbow': 18,
    'Left Wrist': 20, 'Neck (LSP)': 47, 'Top of Head (LSP)': 48, 'Pelvis (MPII)': 49, 'Thorax (MPII)': 50,
    'Spine (H36M)': 51, 'Jaw (H36M)': 52, 'Head (H36M)': 53, 'Nose': 24, 'Left Eye': 26, 'Right Eye': 25,
    'Left Ear': 28, 'Right Ear': 27}
# smplx 0.1.13 vertex_ids['smplh'] in the order VertexJointSelector concatenates them (face, feet,
# left-hand tips, right-hand tips): joints 24..44 of smplx.SMPL.forward.
SMPLX_SMPL_EXTRA_VERTEX_IDS = [
    332, 6260, 2800, 4071, 583,                       # nose, reye, leye, rear, lear
    3216, 3226, 3387, 6617, 6624, 6787,               # LBigToe, LSmallToe, LHeel, RBigToe, RSmallToe, RHeel
    2746, 2319, 2445, 2556, 2673,                     # lthumb, lindex, lmiddle, lring, lpinky
    6191, 5782, 5905, 6016, 6133]                     # rthumb, rindex, rmiddle, rring, rpinky
DEFAULT_JOINT_REGRESSOR_TRAIN_EXTRA = 'data/essentials/spin/J_regressor_extra.npy'      # configs/config.py:77


def reference_config():
    """The reference checkout's ``configs.config`` when it is importable (asset paths), else None."""
    try:
        return importlib.import_module('configs.config')
    except ImportError:
        return None


def reference_constants():
    """``data.essentials.constants`` when the licensed data folder is importable, else None."""
    try:
        return importlib.import_module('data.essentials.constants')
    except ImportError:
        return None


def spin_joint_map():
    """[constants.JOINT_MAP[n] for n in constants.JOINT_NAMES] (tuch/models/smpl.py:38), from the data
    folder when present, else from the tables above."""
    c = reference_constants()
    if c is not None and hasattr(c, 'JOINT_MAP') and hasattr(c, 'JOINT_NAMES'):
        return [c.JOINT_MAP[n] for n in c.JOINT_NAMES]
    return [SPIN_JOINT_MAP[n] for n in SPIN_JOINT_NAMES]


def spin_joint_ids():
    c = reference_constants()
    if c is not None and hasattr(c, 'JOINT_IDS'):
        return dict(c.JOINT_IDS)
    return {n: i for i, n in enumerate(SPIN_JOINT_NAMES)}


class _ChumpyShim:
    """Stands in for chumpy.ch.Ch when the official SMPL pickle is read without chumpy installed: keeps
    the pickled state; the array is its ``x`` (leaf Ch) entry."""

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {'x': state})

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self.__dict__['x'])
        return a.astype(dtype) if dtype is not None else a


class _SmplUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split('.')[0] == 'chumpy':
            return _ChumpyShim
        return super().find_class(module, name)


def _read_pickle(path):
    with open(path, 'rb') as f:
        raw = f.read()
    try:
        return pickle.loads(raw, encoding='latin1')
    except ModuleNotFoundError as exc:        # chumpy objects inside the official model files
        if 'chumpy' not in str(exc):
            raise
        return _SmplUnpickler(io.BytesIO(raw), encoding='latin1').load()


ModelOutput = namedtuple('ModelOutput',
                         ['vertices', 'joints', 'full_pose', 'betas', 'global_orient', 'body_pose'])


def _load_model_dir(model_dir, gender='neutral'):
    """SMPL arrays from ``model_dir`` (a file, or a directory holding SMPL_{GENDER}.pkl/.npz)."""
    path = model_dir
    if os.path.isdir(path):
        for ext in ('pkl', 'npz'):             # smplx: os.path.join(model_path, 'SMPL_{}.pkl'.format(gender.upper()))
            cand = os.path.join(path, 'SMPL_{}.{}'.format(gender.upper(), ext))
            if os.path.exists(cand):
                path = cand
                break
        else:
            raise FileNotFoundError('no SMPL_%s.pkl / .npz under %s' % (gender.upper(), path))
    if path.endswith('.npz'):
        d = dict(np.load(path, allow_pickle=True))
    else:
        d = _read_pickle(path)
    parents = np.asarray(d['kintree_table'])[0].astype(np.int64)
    parents[0] = -1
    num_verts = np.asarray(d['v_template']).shape[0]
    posedirs = np.asarray(d['posedirs'])
    if posedirs.ndim == 3:          # official layout [V,3,207] -> smplx's [207, V*3]
        posedirs = posedirs.reshape(num_verts * 3, -1).T
    return dict(v_template=np.asarray(d['v_template']), shapedirs=np.asarray(d['shapedirs'])[:, :, :10],
                posedirs=posedirs, J_regressor=np.asarray(d['J_regressor'].todense()
                                                          if hasattr(d['J_regressor'], 'todense')
                                                          else d['J_regressor']),
                lbs_weights=np.asarray(d['weights']), parents=parents, faces=np.asarray(d['f']),
                # not an official key: synthetic models whose topology is not SMPL's carry their picked vertices
                **({'extra_vertex_ids': np.asarray(d['extra_vertex_ids'])} if 'extra_vertex_ids' in d else {}))


class SMPL(nn.Module):
    NUM_JOINTS = 23
    NUM_BODY_JOINTS = 23

    def __init__(self, model_dir=None, batch_size=1, create_transl=False, gender='neutral',
                 model_data=None, J_regressor_extra=None, joint_map=None, extra_vertex_ids=None,
                 dtype=torch.float32, **kwargs):
        super().__init__()
        self.batch_size = batch_size
        self.dtype = dtype
        if model_data is None:
            model_data = _load_model_dir(model_dir, gender)
        get = (lambda k: getattr(model_data, k)) if not isinstance(model_data, dict) else model_data.__getitem__
        has = (lambda k: hasattr(model_data, k)) if not isinstance(model_data, dict) else model_data.__contains__
        f32 = lambda a: torch.tensor(np.asarray(a), dtype=dtype)
        self.faces = np.asarray(get('faces'))
        self.register_buffer('faces_tensor', torch.tensor(self.faces.astype(np.int64)))
        self.register_buffer('v_template', f32(get('v_template')))
        self.register_buffer('shapedirs', f32(get('shapedirs')))
        self.register_buffer('posedirs', f32(get('posedirs')))
        self.register_buffer('J_regressor', f32(get('J_regressor')))
        self.register_buffer('lbs_weights', f32(get('lbs_weights')))
        self.register_buffer('parents', torch.tensor(np.asarray(get('parents')).astype(np.int64)))
        if extra_vertex_ids is None and has('extra_vertex_ids'):
            extra_vertex_ids = get('extra_vertex_ids')
        if J_regressor_extra is None and has('J_regressor_extra'):
            J_regressor_extra = get('J_regressor_extra')
        if joint_map is None and has('joint_map'):
            joint_map = get('joint_map')
        if extra_vertex_ids is None:           # smplx VertexJointSelector (vertex_ids['smplh'])
            extra_vertex_ids = SMPLX_SMPL_EXTRA_VERTEX_IDS
            if max(extra_vertex_ids) >= self.v_template.shape[0]:
                raise ValueError('the smplx vertex-id table needs the 6890-vertex SMPL topology; pass extra_vertex_ids')
        if J_regressor_extra is None:          # tuch/models/smpl.py:39
            cfg = reference_config()
            path = getattr(cfg, 'JOINT_REGRESSOR_TRAIN_EXTRA', DEFAULT_JOINT_REGRESSOR_TRAIN_EXTRA)
            J_regressor_extra = np.load(path)
        if joint_map is None:                  # tuch/models/smpl.py:38,42
            joint_map = spin_joint_map()
        self.register_buffer('extra_vertex_ids', torch.tensor(np.asarray(extra_vertex_ids).astype(np.int64)))
        self.register_buffer('J_regressor_extra', f32(J_regressor_extra))      # models/smpl.py:39-41
        self.joint_map = torch.tensor(np.asarray(joint_map).astype(np.int64))  # models/smpl.py:42

    def get_num_verts(self):
        return self.v_template.shape[0]

    def forward(self, betas=None, body_pose=None, global_orient=None, pose2rot=True,
                return_full_pose=False, **kwargs):
        """Reference: tuch/models/smpl.py:44-56 over smplx SMPL.forward."""
        from .. import lbs
        w = 3 if pose2rot else 9
        rows = lambda t, n: t if (t.dim() == 2 and t.shape[1] == n) else t.reshape(-1, n)
        # (a tensor that already has the row shape is passed on AS THE SAME OBJECT: ops._Stage2Tail recognises the pose tensor
        # it shares with this node by identity)
        global_orient_rows = rows(global_orient, w)
        body_pose_rows = rows(body_pose, 23 * w)
        if betas.shape[0] != body_pose_rows.shape[0]:
            betas = betas.expand(body_pose_rows.shape[0], -1)
        # the kernels read the two pose tensors where they are; the concatenation of models/smpl.py:44-47 only on request
        vertices, joints = lbs.smpl_forward_split(self, betas, global_orient_rows, body_pose_rows, pose2rot)
        full_pose = None
        if return_full_pose:
            if pose2rot:
                full_pose = torch.cat([global_orient_rows, body_pose_rows], dim=1)
            else:
                full_pose = torch.cat([global_orient.reshape(-1, 1, 3, 3), body_pose.reshape(-1, 23, 3, 3)], dim=1)
        return ModelOutput(vertices=vertices, joints=joints, betas=betas, global_orient=global_orient,
                           body_pose=body_pose, full_pose=full_pose)
