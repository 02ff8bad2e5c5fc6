"""CPU oracle for the non-contact terms of the SMPLify-DC objective and its per-body composition --
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  NumPy float32; every function cites the
reference lines it restates.  Pinned by tests/test_oracle_golden.py against values produced by the
reference's own functions (tests/golden/contact_*.npz: projected_joints, gmof_values, prior_values,
body_fitting_reprojection, smplify_*_full_loss).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np

from . import contact as oc


def perspective_projection(points, translation, focal_length, camera_center):
    """tuch/utils/geometry.py:83-111 with the identity rotation every caller passes:
    p = (X + t) / (X + t)_z ; uv = f p_xy + c.   [B,N,3], [B,3], scalar, [B,2] -> [B,N,2]."""
    p = np.asarray(points, np.float32) + np.asarray(translation, np.float32)[:, None, :]
    p = p / p[:, :, 2:3]
    f = np.float32(focal_length)
    return (f * p[:, :, :2] + np.asarray(camera_center, np.float32)[:, None, :]).astype(np.float32)


def gmof(x, sigma):
    """tuch/smplify/losses.py:25-32: sigma^2 x^2 / (sigma^2 + x^2)."""
    x2 = np.asarray(x, np.float32) ** 2
    s2 = np.float32(sigma) ** 2
    return (s2 * x2) / (s2 + x2)


def reprojection(joints, camera_t, camera_center, joints_2d, joints_conf, focal_length=5000., sigma=100.):
    """losses.py:56-61 / :141-146 / :175-180: conf^2 * sum_xy gmof(proj - j2d) -> [B,49]."""
    proj = perspective_projection(joints, camera_t, focal_length, camera_center)
    return (np.asarray(joints_conf, np.float32) ** 2) * gmof(proj - np.asarray(joints_2d, np.float32), sigma).sum(-1)


def merged_prior(pose, gmm: Dict[str, np.ndarray]):
    """tuch/smplify/prior.py:88-96 (weights folded with the normalisers, in float64 as numpy does there) and
    :117-132 (min over the components of 0.5 maha - log w').  pose [B,69] -> [B]."""
    means = np.asarray(gmm['means'], np.float64).astype(np.float32)
    covs64 = np.asarray(gmm['covars'], np.float64)
    prec = np.stack([np.linalg.inv(c) for c in covs64.astype(np.float32)]).astype(np.float32)
    root_det = np.sqrt(np.array([np.linalg.det(c) for c in covs64]))
    w = np.asarray(gmm['weights'], np.float64) / ((2 * np.pi) ** (69 / 2.) * (root_det / root_det.min()))
    w = w.astype(np.float32)
    diff = np.asarray(pose, np.float32)[:, None, :] - means[None]
    maha = np.einsum('bmi,mij,bmj->bm', diff, prec, diff).astype(np.float32)
    return (np.float32(0.5) * maha - np.log(w)[None]).min(1)


def stage2_objective(verts, joints, body_pose, faces, geomask, euclthres, camera_t, camera_center, joints_2d,
                     joints_conf, gmm, segments: Optional[Sequence] = None,
                     region_pairs_per_body: Optional[Sequence] = None, ignore: Optional[np.ndarray] = None,
                     focal_length=5000., sigma=100., contact_loss_weight=1000.):
    """contact_fitting_loss, tuch/smplify/losses.py:34-123, for a batch: per body
    sum_j conf^2 gmof + 10 contact + prior + clw r2r; returns (total, per_body[B], per-body dicts).
    region_pairs_per_body[b]: list of (verts1_idxs, verts2_idxs) of the annotated pairs of body b, or None
    when has_discrete_contact[b] is false (losses.py:108-117)."""
    batch = verts.shape[0]
    rep = reprojection(joints, camera_t, camera_center, joints_2d, joints_conf, focal_length, sigma).sum(1)
    prior = merged_prior(body_pose, gmm) if gmm is not None else np.zeros(batch, np.float32)
    per_body = np.zeros(batch, np.float64)
    parts = []
    for b in range(batch):
        r = None
        if ignore is None or not ignore[b]:                                      # losses.py:73
            rp = region_pairs_per_body[b] if region_pairs_per_body is not None else None
            r = oc.smplify_contact_body(verts[b], faces, geomask, euclthres, segments, rp)
        contact = r['contact'] if r is not None else 0.0
        r2r = r['r2r'] if r is not None else 0.0
        per_body[b] = float(rep[b]) + 10.0 * contact + float(prior[b]) + contact_loss_weight * r2r   # :120-121
        parts.append(dict(reprojection=float(rep[b]), prior=float(prior[b]), contact=contact, r2r=r2r))
    return float(per_body.sum()), per_body, parts
