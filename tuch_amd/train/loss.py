"""Drop-in for the reference's ``tuch/train/loss.py``: ``RegressorLoss`` with the L_P / L_C
contact loss on the HIP kernels; the SPIN terms (keypoints, shape, pose/betas regression,
camera) are small torch reductions as in the reference.

The reference's constructor call works verbatim (train.py:79-87):

    RegressorLoss(options=..., device=..., num_verts=..., faces=face_tensor, geodistssmpl=...,
                  geothres=config.geothres, face_tensor=face_tensor)

and then loads what the reference loads: the HD regressor files from config.HD_MODEL_DIR
(loss.py:81-88) and the body segments from config.SEGMENT_DIR + segm_utils (loss.py:91).  A missing
asset raises -- the segment filter is never skipped silently.  Because the licensed assets do not
ship, the same data may be injected instead: ``segments`` (a BatchBodySegment), ``hd_regressor`` =
(idx [N,3], weights [N,3]) -- the three non-zeros of every row of smpl_neutral_hd_vert_regressor.npy
-- and ``hd_faces`` = faces_vert_is_sampled_from.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import dist as tdist, ops
from ..utils.geometry import batch_rodrigues


def batch_face_normals(triangles):
    """Unit normals of [B,F,3,3] triangles (reference: loss.py:30-41)."""
    n = torch.cross(triangles[:, :, 1] - triangles[:, :, 0], triangles[:, :, 2] - triangles[:, :, 0], dim=2)
    return n / torch.norm(n, 2, dim=2, keepdim=True)


class RegressorLoss(nn.Module):
    def __init__(self, options, device, num_verts, faces, geodistssmpl, geothres=0.2, euclthres=0.02,
                 face_tensor=None, use_hd=True, segments=None, hd_regressor=None, hd_faces=None,
                 hd_model_dir=None, global_mean=False):
        super().__init__()
        self.device = device
        # contact_loss is a mean over the valid bodies (loss.py:317).  Default: over THIS process's bodies, like every
        # other term of forward() -- under DistributedDataParallel (gradients averaged over ranks) all terms then share
        # one normalisation.  global_mean=True divides the local sum by the all-reduced count of valid bodies instead
        # (one float over RCCL on the calling stream): the ranks' values then SUM to the single-process mean and every
        # body's gradient is the single-process gradient, which is what a trainer that SUMS gradients wants (and what
        # makes ranks with few valid bodies weigh less); loss_dict['loss_contact'] is then the rank's share.
        self.global_mean = bool(global_mean)
        self.options = options
        self.criterion_shape = nn.L1Loss().to(self.device)
        self.criterion_keypoints = nn.MSELoss(reduction='none').to(self.device)
        self.criterion_regr = nn.MSELoss().to(self.device)
        self.faces = faces
        self.nv = num_verts
        self.geodistssmpl = geodistssmpl
        self.geothres = geothres
        self.geomask = geodistssmpl > geothres                       # loss.py:71 (strict >)
        self.euclthres = euclthres
        self.face_tensor = face_tensor
        self.use_hd = use_hd
        if use_hd:
            if hd_regressor is None:                                          # loss.py:81-87
                from ..assets import load_hd_regressor
                hd_i, hd_wt, hd_faces = load_hd_regressor(hd_model_dir)
                hd_regressor = (hd_i, hd_wt)
            dev = face_tensor.device
            hd_i, hd_wt, hd_f = np.asarray(hd_regressor[0]), np.asarray(hd_regressor[1]), np.asarray(hd_faces)
            # attributes of the reference (loss.py:83-88), in the reference's point order
            self.hd_idx = torch.as_tensor(hd_i, dtype=torch.long, device=dev)
            self.hd_w = torch.as_tensor(hd_wt, dtype=torch.float32, device=dev).contiguous()
            self.geovec = torch.as_tensor(hd_f, dtype=torch.long, device=dev)
            self.geovec_verts = self.face_tensor[0][self.geovec][:, 0]            # loss.py:88
        if segments is None:                                                  # loss.py:91: always built
            from ..utils.segmentation import BatchBodySegment, reference_segment_names
            segments = BatchBodySegment(reference_segment_names(), self.face_tensor[0])
        self.segments = segments
        self._model = ops.ContactModel(face_tensor[0], self.geomask, segments.tables(), device=face_tensor.device)
        if use_hd:
            # device tables of the fused HD branch (csrc/hd_contact.hip); the library keeps the points sorted by the
            # surface patch of their face (the loss is a sum over them, their order is free)
            self._hd = ops.HDModel(self._model, hd_i, hd_wt, hd_f)

    # ------------------------------------------------------------------ contact (loss.py:240-317)
    def contact_loss(self, pred_vertices, valid_fit):
        """mean over the valid bodies of  sum_exterior 0.005 tanh^2(d/0.005) + sum_interior tanh^2(d/0.04)
        with d the distance to the nearest geodesically-far point, on the SMPL vertices
        (use_hd=False) or on the HD points resampled around contact / interior (use_hd=True)."""
        model = self._model
        valid = valid_fit.bool()
        valid_u8 = ops.as_u8(valid)
        exterior, min_d2, partner, _ = model.exterior_and_partner(pred_vertices, apply_segments=True)   # :264-266
        # loss.py:317: mean over the valid bodies -- of this process (one launch: ops.valid_mean), or (global_mean) of all
        # ranks (the count is summed over the process group: torch)
        fused = not self.global_mean and pred_vertices.is_cuda
        if not fused:
            n_valid = tdist.global_count(valid.sum()) if self.global_mean else valid.sum().to(torch.float32)
        if not self.use_hd:
            if fused:
                return ops.contact_terms_mean(pred_vertices, partner, exterior, valid_u8, ops.MODE_TRAIN, self.euclthres)
            per_body, _ = ops.contact_terms(pred_vertices, partner, exterior, valid_u8, ops.MODE_TRAIN,
                                            self.euclthres)
            return per_body.sum() / n_valid
        # HD branch, loss.py:274-315: one fixed sequence of kernels, no host synchronisation (hipGraph-capturable)
        terms = self._hd.contact_terms(pred_vertices, exterior, min_d2, partner, valid_u8, self.euclthres)
        if fused:
            return ops.valid_mean(terms, valid_u8)
        return terms.sum() / n_valid

    # ---------------------------------------------------------------------------- SPIN terms
    def forward(self, pred_rotmat, pred_betas, opt_pose, opt_betas, pred_keypoints_2d, gt_keypoints_2d,
                pred_joints, gt_joints, has_pose_3d, pred_vertices, opt_vertices, pred_camera, valid_fit,
                valid_fit_shape):
        """Reference: loss.py:94-168 -> (total_loss, loss_dict with the same 7 keys)."""
        loss_contact = torch.tensor(0)
        if self.options.contact_loss_weight > 0:
            loss_contact = self.contact_loss(pred_vertices, valid_fit)
        contact_loss = self.options.contact_loss_weight * loss_contact
        loss_regr_pose, loss_regr_betas = self.smpl_losses(pred_rotmat, pred_betas, opt_pose, opt_betas,
                                                           valid_fit, valid_fit_shape)
        loss_keypoints = self.keypoint_loss(pred_keypoints_2d, gt_keypoints_2d,
                                            self.options.openpose_train_weight, self.options.gt_train_weight,
                                            valid_fit)
        loss_keypoints_3d = self.keypoint_3d_loss(pred_joints, gt_joints, has_pose_3d)
        loss_shape = self.shape_loss(pred_vertices, opt_vertices, valid_fit)
        cam_loss = ((torch.exp(-pred_camera[:, 0] * 10)) ** 2).mean()
        o = self.options
        spin_loss = o.shape_loss_weight * loss_shape + o.keypoint_loss_weight * loss_keypoints + \
            o.keypoint_loss_weight * loss_keypoints_3d + o.pose_loss_weight * loss_regr_pose + \
            o.beta_loss_weight * loss_regr_betas + cam_loss
        loss_dict = {'loss_shape': loss_shape, 'loss_keypoints': loss_keypoints,
                     'loss_keypoints_3d': loss_keypoints_3d, 'loss_regr_pose': loss_regr_pose,
                     'loss_regr_betas': loss_regr_betas, 'loss_cam': cam_loss, 'loss_contact': loss_contact}
        return spin_loss + contact_loss, loss_dict

    def _zero(self):
        return torch.zeros(1, dtype=torch.float32, device=self.device)

    # The SPIN terms below select samples with boolean masks (loss.py:172-238: ``x[mask]``).  Boolean indexing costs a
    # device -> host synchronisation per use (nonzero): 15 per training step.  They are written as masked reductions
    # instead -- the same means (sum over the selected samples / their element count; 0 where the reference returns 0
    # for an empty selection, NaN where it takes the mean of nothing), no host round trip.
    @staticmethod
    def _selected_mean(per_sample, mask, elems_per_sample, empty_is_zero):
        """mean over the elements of the selected samples; per_sample [B] = sum over a sample's elements."""
        m = mask.reshape(-1).bool()
        n = m.sum().to(per_sample.dtype)
        total = torch.where(m, per_sample, torch.zeros_like(per_sample)).sum()
        mean = total / (n * elems_per_sample)
        return torch.where(n > 0, mean, torch.zeros_like(mean)) if empty_is_zero else mean

    def keypoint_loss(self, pred_keypoints_2d, gt_keypoints_2d, openpose_weight, gt_weight, valid_fit=None):
        """Confidence-weighted 2D reprojection MSE (loss.py:172-183)."""
        conf = gt_keypoints_2d[:, :, -1].unsqueeze(-1).clone()
        conf[:, :25] *= openpose_weight
        conf[:, 25:] *= gt_weight
        loss = (conf * self.criterion_keypoints(pred_keypoints_2d, gt_keypoints_2d[:, :, :-1])).mean(axis=(1, 2))
        return self._selected_mean(loss, valid_fit, 1, empty_is_zero=False)

    def keypoint_3d_loss(self, pred_keypoints_3d, gt_keypoints_3d, has_pose_3d):
        """Pelvis-centred 3D keypoint MSE on the 24 ground-truth joints (loss.py:185-204)."""
        pred = pred_keypoints_3d[:, 25:, :]
        conf = gt_keypoints_3d[:, :, -1].unsqueeze(-1)
        gt = gt_keypoints_3d[:, :, :-1]
        gt = gt - ((gt[:, 2, :] + gt[:, 3, :]) / 2)[:, None, :]
        pred = pred - ((pred[:, 2, :] + pred[:, 3, :]) / 2)[:, None, :]
        per = (conf * self.criterion_keypoints(pred, gt)).sum(dim=(1, 2))
        return self._selected_mean(per, has_pose_3d == 1, gt.shape[1] * gt.shape[2], empty_is_zero=True)

    def shape_loss(self, pred_vertices, gt_vertices, has_smpl):
        """Per-vertex L1 where SMPL fits exist (loss.py:206-215)."""
        per = (pred_vertices - gt_vertices).abs().sum(dim=(1, 2))
        return self._selected_mean(per, has_smpl == 1, pred_vertices.shape[1] * pred_vertices.shape[2], empty_is_zero=True)

    def smpl_losses(self, pred_rotmat, pred_betas, gt_pose, gt_betas, has_smpl_pose, has_smpl_shape):
        """MSE on rotation matrices and betas (loss.py:217-238)."""
        gr = batch_rodrigues(gt_pose.view(-1, 3)).view(-1, 24, 3, 3)
        per_pose = ((pred_rotmat - gr) ** 2).sum(dim=(1, 2, 3))
        loss_pose = self._selected_mean(per_pose, has_smpl_pose == 1, 24 * 9, empty_is_zero=True)
        per_betas = ((pred_betas - gt_betas) ** 2).sum(dim=1)
        loss_betas = self._selected_mean(per_betas, has_smpl_shape == 1, pred_betas.shape[1], empty_is_zero=True)
        return loss_pose, loss_betas
