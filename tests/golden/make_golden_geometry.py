#!/usr/bin/env python3
"""Golden vectors for the caller-side geometry glue, produced by the REFERENCE's own functions
(/root/reference/tuch/utils/geometry.py, imported, never copied) on synthetic inputs:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_geometry.py

Writes tests/golden/geometry.npz (inputs next to expected outputs).
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(1, '/root/reference')

import numpy as np
import torch

from tuch.utils import geometry as ref                      # noqa: E402

rng = np.random.default_rng(77)
B = 24
S = (rng.standard_normal((B, 49, 3)) * 0.4).astype(np.float32)
t_true = np.stack([rng.uniform(-0.5, 0.5, B), rng.uniform(-0.5, 0.5, B), rng.uniform(4, 40, B)], 1).astype(np.float32)
focal, img = 5000.0, 224.0
cam = S + t_true[:, None]
uv = focal * cam[:, :, :2] / cam[:, :, 2:3] + img / 2 + rng.standard_normal((B, 49, 2)) * 1.5
conf = rng.uniform(0.0, 1.0, (B, 49)).astype(np.float32)
conf[rng.uniform(size=conf.shape) < 0.2] = 0.0
conf[3, :25] = 0.0            # a sample without confident OpenPose joints -> zeros (if it uses them)
conf[5, 25:] = 0.0
kp = np.concatenate([uv.astype(np.float32), conf[:, :, None]], 2).astype(np.float32)
anno = rng.uniform(size=B) < 0.5
anno[3], anno[5] = False, True
trans = ref.estimate_translation(torch.tensor(S), torch.tensor(kp), focal_length=focal, img_size=img,
                                 has_2d_kp_anno=torch.tensor(anno)).numpy()
trans_1000 = ref.estimate_translation(torch.tensor(S), torch.tensor(kp), focal_length=1000.0, img_size=256.0,
                                      has_2d_kp_anno=torch.tensor(anno)).numpy()
x6 = rng.standard_normal((B * 24, 6)).astype(np.float32)
rot6d = ref.rot6d_to_rotmat(torch.tensor(x6)).numpy()
aa = (rng.standard_normal((64, 3)) * 1.2).astype(np.float32)
aa[0] = 0.0
rod = ref.batch_rodrigues(torch.tensor(aa)).numpy()
np.savez_compressed(os.path.join(HERE, 'geometry.npz'), S=S, kp=kp, anno=anno, trans=trans, trans_f1000=trans_1000,
                    x6=x6, rot6d=rot6d, aa=aa, rodrigues=rod)
print('wrote geometry.npz', trans[:6])
