"""Drop-in for the reference's ``tuch/smplify/losses.py`` (same function names and argument
lists).  The contact part of ``contact_fitting_loss`` -- the per-body Python loop of
losses.py:74-117 -- is one batched pass over the HIP kernels; reprojection, GMoF and the
priors are [B,49]-sized torch ops (K8).
"""
from __future__ import annotations

import os

from typing import Dict, Optional

import numpy as np
import torch

from .. import ops
from ..utils.geometry import perspective_projection

# one autograd node for everything behind the body model (ops._Stage2Tail) instead of separate nodes; read from the
# environment once, at import (TUCH_FUSED_TAIL=0: A/B measurements; the tests flip the attribute)
FUSED_TAIL = os.environ.get('TUCH_FUSED_TAIL', '1') != '0'
# camera_fitting_loss (the stage-1 objective) as one launch on HIP devices (ops._Stage1Objective); 0: the torch-op form
STAGE1_FUSED = os.environ.get('TUCH_STAGE1_FUSED', '1') != '0'

_MODEL_CACHE: Dict[tuple, tuple] = {}
_ANGLE_SIGNS: Dict[tuple, tuple] = {}


def gmof(x, sigma):
    """Geman-McClure robustifier sigma^2 x^2 / (sigma^2 + x^2) (reference: losses.py:25-32)."""
    x2, s2 = x ** 2, sigma ** 2
    return (s2 * x2) / (s2 + x2)


def contact_model_for(geomask, face_tensor, segments=None, cdict=None, device=None) -> ops.ContactModel:
    """The device-side constants for one (geomask, faces, segments, regions) combination, built
    on first use and cached: the reference passes these objects on every call
    (smplifydc.py:162-179) and keeps them alive for the whole run."""
    faces = face_tensor[0] if face_tensor.dim() == 3 else face_tensor
    key = (geomask.data_ptr() if torch.is_tensor(geomask) else id(geomask), faces.data_ptr(),
           id(segments), id(cdict))
    entry = _MODEL_CACHE.get(key)
    model = entry[0] if entry is not None else None
    if model is None:
        seg_tables = segments.tables() if segments is not None else None
        regions = pairs = None
        # SMPLifyDC.__call__ defaults contactlist=[] (smplifydc.py:70): anything that is not a dict with
        # region pairs means "no region term" (the reference only reads cdict for annotated bodies)
        if isinstance(cdict, dict) and len(cdict.get('classes', [])) > 0:
            regions, pairs = ops.region_tables(cdict)
        model = ops.ContactModel(faces, geomask, seg_tables, regions, pairs,
                                 device=device if device is not None else faces.device)
        # keep the keyed objects alive with the entry: a freed tensor's address / a dead object's id can
        # be reused, which would turn the key into a stale hit
        while len(_MODEL_CACHE) >= 32:
            _MODEL_CACHE.pop(next(iter(_MODEL_CACHE)))
        _MODEL_CACHE[key] = (model, geomask, face_tensor, segments, cdict)
    return model


def _reprojection(model_joints, camera_t, camera_center, joints_2d, joints_conf, focal_length, sigma):
    batch = model_joints.shape[0]
    eye = torch.eye(3, device=model_joints.device).unsqueeze(0).expand(batch, -1, -1)
    proj = perspective_projection(model_joints, eye, camera_t, focal_length, camera_center)
    return (joints_conf ** 2) * gmof(proj - joints_2d, sigma).sum(dim=-1)


def contact_fitting_loss(body_pose, global_orient, body_pose_loop1, opt_global_orient_smplifyloop1,
                         betas, model_joints, geomask, euclthres,
                         camera_t, camera_center,
                         joints_2d, joints_conf, pose_prior,
                         cdict, gt_contact,
                         ignore_idxs,
                         has_discrete_contact,
                         verts, face_tensor=None,
                         device=None,
                         focal_length=5000, sigma=100, pose_prior_weight=1.0,
                         shape_prior_weight=1.0, angle_prior_weight=1.0,
                         contact_loss_weight=1000, output='sum',
                         segments=None):
    """SMPLify-DC stage-2 objective (reference: losses.py:34-123):

        sum_b [ sum_j conf^2 gmof(proj - j2d) + 10 * contact_b + ppw^2 prior_b + clw * r2r_b ]

    contact_b = sum_{interior} tanh^2(d/0.04) + sum_{exterior, d<euclthres} 0.005 tanh^2(d/0.005)
    with d_i the distance to the nearest geodesically-far vertex; r2r_b = sum over the annotated
    region pairs of the (geodesically masked) minimum squared distance between the regions.
    Bodies in ``ignore_idxs`` get neither contact term.  ``device`` is accepted for signature
    compatibility; the computation runs where ``verts`` lives.
    """
    model = contact_model_for(geomask, face_tensor, segments, cdict, device=verts.device)
    valid = ops.cached_derived((ignore_idxs,), lambda: (~ignore_idxs).to(torch.uint8).contiguous())
    select = None
    if model.num_pairs > 0 and gt_contact is not None and gt_contact[0] is not None:
        select = ops.cached_derived(
            (gt_contact[0], has_discrete_contact, ignore_idxs),
            lambda: ((gt_contact[0] == 1) & has_discrete_contact.bool()[:, None]
                     & (~ignore_idxs)[:, None]).to(torch.uint8).contiguous())
    return stage2_objective(model, valid, select, body_pose, betas, model_joints, euclthres, camera_t, camera_center,
                            joints_2d, joints_conf, pose_prior, verts, focal_length, sigma, pose_prior_weight,
                            contact_loss_weight, apply_segments=segments is not None)


def stage2_objective(model, valid, select, body_pose, betas, model_joints, euclthres, camera_t, camera_center, joints_2d,
                     joints_conf, pose_prior, verts, focal_length=5000, sigma=100, pose_prior_weight=1.0,
                     contact_loss_weight=1000, apply_segments=True):
    """The body of contact_fitting_loss on prepared constants: ``model`` (ops.ContactModel), ``valid`` [B] u8 = bodies
    that get contact terms (~ignore_idxs), ``select`` [B,P] u8 = annotated region pairs of the bodies with discrete
    contact (or None)."""
    from .prior import MaxMixturePrior
    fused = (isinstance(pose_prior, MaxMixturePrior) and pose_prior.use_merged and verts.is_cuda
             and not torch.is_tensor(focal_length) and body_pose.shape[1] == 69
             and pose_prior.means.dtype == torch.float32 and pose_prior.means.is_cuda
             and model_joints.dtype == torch.float32 and body_pose.dtype == torch.float32)

    if fused and camera_t.shape == (body_pose.shape[0], 3) and FUSED_TAIL:
        # one autograd node for everything behind the body model (ops._Stage2Tail): the same kernels, less glue
        return ops.smplify_stage2_tail(
            verts, model_joints, camera_t, body_pose, model, valid,
            select.to(torch.uint8).contiguous() if select is not None else None,
            camera_center=camera_center, joints_2d=joints_2d, joints_conf=joints_conf, means=pose_prior.means,
            precisions=pose_prior.precisions, log_weights=pose_prior.log_nll_weights, focal=focal_length, sigma=sigma,
            prior_scale=pose_prior_weight ** 2, euclthres=euclthres, contact_scale=10.0, r2r_scale=contact_loss_weight,
            apply_segments=apply_segments)

    def beside_the_walk():
        """Everything that does not need the inside test: region pairs (losses.py:107-117) and, fused into one
        kernel (K8), projection + gmof + GMM prior (losses.py:56-64).  Runs on the second stream."""
        r = model.region_pair_min(verts, select=select, masked=True)[0] if select is not None else None
        sm = ops.smplify_small_terms(model_joints, camera_t, body_pose, camera_center, joints_2d, joints_conf,
                                     pose_prior.means, pose_prior.precisions, pose_prior.log_nll_weights,
                                     focal_length, sigma, pose_prior_weight ** 2) if fused else None
        return r, sm
    # losses.py:79-89 (inside test) and losses.py:76-78,92-93 (nearest geodesically-far vertex)
    exterior, _, partner, (r2r, small) = model.exterior_and_partner(verts, apply_segments=apply_segments,
                                                                    also=beside_the_walk, iterative=True)
    contact_loss, contact_terms = ops.contact_terms(verts, partner, exterior, valid, ops.MODE_SMPLIFY, euclthres)
    if fused:      # objective assembled in one deterministic reduction
        return ops.smplify_objective(small, contact_terms, r2r, 10.0, contact_loss_weight)   # losses.py:120-123
    reprojection_sum = _reprojection(model_joints, camera_t, camera_center, joints_2d, joints_conf,
                                     focal_length, sigma).sum(dim=-1)
    pose_prior_loss = (pose_prior_weight ** 2) * pose_prior(body_pose, betas)
    r2r_loss = r2r.sum(dim=1) if r2r is not None else torch.zeros_like(contact_loss)
    total_loss = reprojection_sum + 10 * contact_loss \
        + pose_prior_loss + contact_loss_weight * r2r_loss
    return total_loss.sum()


def camera_fitting_loss(smpl_output, camera_t, camera_t_est, camera_center, joints_2d, joints_conf,
                        focal_length=5000, depth_loss_weight=100, sigma=100, shape_prior_weight=0.0):
    """Stage-1 objective for camera translation / betas (reference: losses.py:125-152).  On a HIP device the whole
    objective and its gradients are one kernel launch (ops.smplify_stage1_objective); the torch-op form below is the
    same arithmetic for other devices / dtypes."""
    joints = smpl_output.joints
    if joints.is_cuda and joints.dtype == torch.float32 and STAGE1_FUSED:
        return ops.smplify_stage1_objective(joints, camera_t, smpl_output.betas if shape_prior_weight != 0 else None,
                                            camera_t_est, camera_center, joints_2d, joints_conf, focal_length, sigma,
                                            depth_loss_weight, shape_prior_weight)
    reprojection_loss = _reprojection(smpl_output.joints, camera_t, camera_center, joints_2d, joints_conf,
                                      focal_length, sigma)
    depth_loss = (depth_loss_weight ** 2) * (camera_t[:, 2] - camera_t_est[:, 2]) ** 2
    shape_prior_loss = (shape_prior_weight ** 2) * (smpl_output.betas ** 2).sum(dim=-1)
    return (reprojection_loss.sum(dim=-1) + depth_loss + shape_prior_loss).sum()


def angle_prior(pose):
    """Exponential penalty on unnatural knee / elbow bending (reference: losses.py:155-162;
    indices are into the 69-D body pose, hence the -3)."""
    key = (pose.device, pose.dtype)
    if key not in _ANGLE_SIGNS:       # built once per device: a host->device copy cannot be graph-captured
        _ANGLE_SIGNS[key] = (torch.tensor([1., -1., -1., -1.], device=pose.device, dtype=pose.dtype),
                             torch.tensor([55 - 3, 58 - 3, 12 - 3, 15 - 3], device=pose.device))
    signs, idx = _ANGLE_SIGNS[key]
    return torch.exp(pose.index_select(1, idx) * signs) ** 2


def body_fitting_loss(body_pose, betas, model_joints, camera_t, camera_center,
                      joints_2d, joints_conf, pose_prior,
                      focal_length=5000, sigma=100, pose_prior_weight=4.78,
                      shape_prior_weight=5, angle_prior_weight=15.2,
                      output='sum'):
    """SPIN's SMPLify objective, kept for the no-contact branch and for the final
    per-joint reprojection report (reference: losses.py:164-198)."""
    reprojection_loss = _reprojection(model_joints, camera_t, camera_center, joints_2d, joints_conf,
                                      focal_length, sigma)
    if output == 'reprojection':
        return reprojection_loss
    total = reprojection_loss.sum(dim=-1) \
        + (pose_prior_weight ** 2) * pose_prior(body_pose, betas) \
        + (angle_prior_weight ** 2) * angle_prior(body_pose).sum(dim=-1) \
        + (shape_prior_weight ** 2) * (betas ** 2).sum(dim=-1)
    return total.sum()
