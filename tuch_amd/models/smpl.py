"""Drop-in for the reference's ``tuch/models/smpl.py``: SMPL linear blend skinning with 21
picked vertices + 9 extra regressed joints, re-mapped to the 49 joints the losses use.

The reference subclasses ``smplx.SMPL`` (third-party, absent here) and loads the
licensed model .pkl.  This class takes the same constructor call
``SMPL(model_dir, batch_size=..., create_transl=False)`` plus ``model_data=`` for already
loaded arrays (e.g. tuch_amd.synthetic.SyntheticBody), and returns the same ``ModelOutput``.
"""
from __future__ import annotations

import os
import pickle
from collections import namedtuple

import numpy as np
import torch
import torch.nn as nn

ModelOutput = namedtuple('ModelOutput',
                         ['vertices', 'joints', 'full_pose', 'betas', 'global_orient', 'body_pose'])


def _load_model_dir(model_dir, gender='neutral'):
    """SMPL arrays from ``model_dir`` (a file, or a directory holding SMPL_{GENDER}.pkl/.npz)."""
    path = model_dir
    if os.path.isdir(path):
        for ext in ('npz', 'pkl'):
            cand = os.path.join(path, 'SMPL_{}.{}'.format(gender.upper(), ext))
            if os.path.exists(cand):
                path = cand
                break
    if path.endswith('.npz'):
        d = dict(np.load(path, allow_pickle=True))
    else:
        with open(path, 'rb') as f:
            d = pickle.load(f, encoding='latin1')
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
                lbs_weights=np.asarray(d['weights']), parents=parents, faces=np.asarray(d['f']))


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
        if extra_vertex_ids is None or J_regressor_extra is None or joint_map is None:
            raise ValueError('SMPL needs extra_vertex_ids (21), J_regressor_extra [9,V] and joint_map (49)')
        self.register_buffer('extra_vertex_ids', torch.tensor(np.asarray(extra_vertex_ids).astype(np.int64)))
        self.register_buffer('J_regressor_extra', f32(J_regressor_extra))      # models/smpl.py:39-41
        self.joint_map = torch.tensor(np.asarray(joint_map).astype(np.int64))  # models/smpl.py:42

    def get_num_verts(self):
        return self.v_template.shape[0]

    def forward(self, betas=None, body_pose=None, global_orient=None, pose2rot=True,
                return_full_pose=False, **kwargs):
        """Reference: tuch/models/smpl.py:44-56 over smplx SMPL.forward."""
        from .. import lbs
        if pose2rot:
            full_pose = torch.cat([global_orient.reshape(-1, 3), body_pose.reshape(-1, 69)], dim=1)
        else:
            full_pose = torch.cat([global_orient.reshape(-1, 1, 3, 3), body_pose.reshape(-1, 23, 3, 3)], dim=1)
        if betas.shape[0] != full_pose.shape[0]:
            betas = betas.expand(full_pose.shape[0], -1)
        vertices, joints = lbs.smpl_forward(self, betas, full_pose, pose2rot)
        return ModelOutput(vertices=vertices, joints=joints, betas=betas, global_orient=global_orient,
                           body_pose=body_pose, full_pose=full_pose if return_full_pose else None)
