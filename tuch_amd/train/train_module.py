"""Drop-in for the reference's ``tuch/train/train_module.py``: ``TUCH``, one training step of the regressor with
SMPLify-DC in the loop -- the caller of the hot path (BASELINE configs 4 and 5).

Same constructor and ``forward_train_step(input_batch) -> (loss, losses, output)`` as the reference
(train_module.py:31-66, 105-335).  What runs where:
  * ``contact_from_verts`` (:69-91, "Speed up this function will speed up training loop!") -- one HIP kernel over all
    region pairs instead of a Python loop of three bmm's per pair;
  * ``estimate_translation`` (:171-180, per-sample numpy solves on the host in the reference) and
    ``rotation_matrix_to_angle_axis`` (:208-212) -- HIP kernels, no host round trip;
  * the dictionary of best fits (:142-146, :265) -- on the device (train/fits_dict.py);
  * SMPL, SMPLify-DC and the criterion -- the objects passed in (ours: models/smpl.py, smplify/smplifydc.py,
    train/loss.py);
  * the regressors (HMR / SPIN) are the caller's torch modules and not part of the path.
The per-sample "is the new fit better" bookkeeping (:239-262) is written with ``torch.where`` instead of boolean-mask
assignment, which would synchronise with the host for every tensor.

Extra keyword arguments (the licensed assets do not ship): ``contactlists`` = {'classes', 'csig'} instead of the DSC
pickles, ``fits_dict`` an existing dictionary, ``faces`` when no body model is given
(``TUCH(contactlists=..., faces=..., device=...)`` is enough for ``contact_from_verts``).
"""
from __future__ import annotations

import torch

from .. import ops
from ..utils.geometry import estimate_translation, perspective_projection, rotation_matrix_to_angle_axis


class TUCH():
    def __init__(self, options=None, device=None, datasets=None, bodymodel=None, spin_model=None, regressor=None,
                 optimization=None, criterion=None, geodistssmpl=None, contactlists=None, fits_dict=None, faces=None):
        self.options = options
        self.device = device
        from ..models.smpl import reference_constants
        consts = reference_constants()
        self.focal_length = getattr(consts, 'FOCAL_LENGTH', 5000.) if consts is not None else 5000.
        if datasets is not None:
            self.train_ds, self.val_ds = datasets
        self.modelspin, self.model, self.smplify = spin_model, regressor, optimization
        self.smpl, self.geodistssmpl, self.criterion_cospin = bodymodel, geodistssmpl, criterion
        if fits_dict is None and options is not None and datasets is not None:
            from .fits_dict import FitsDict
            fits_dict = FitsDict(self.options, self.train_ds, device=device)                 # train_module.py:48
        self.fits_dict = fits_dict
        if contactlists is None:                                                                # train_module.py:64-66
            from ..assets import load_contact_regions
            contactlists = load_contact_regions()
        self.contactlists = contactlists
        if faces is None:
            faces = bodymodel.faces
        regions, pairs = ops.region_tables(contactlists)
        self._model = ops.ContactModel(faces, None, None, regions, pairs, device=device)

    # ------------------------------------------------------------------ train_module.py:69-91
    def contact_from_verts(self, verts, mode='regions'):
        """[B,V,3] -> [B,P]: minimum squared distance between the two regions of every pair."""
        if mode != 'regions':
            raise ValueError("only mode='regions' exists in the reference")
        return self._model.region_pair_min(verts)[0]

    # ------------------------------------------------------------------ train_module.py:105-335
    def forward_train_step(self, input_batch):
        o, dev = self.options, self.device
        self.model.train()
        images = input_batch['img']
        batch_size = images.shape[0]
        camera_center = torch.zeros(batch_size, 2, device=dev)
        indices, is_flipped, rot_angle = input_batch['sample_index'], input_batch['is_flipped'], input_batch['rot_angle']
        dataset_name = input_batch['dataset_name']
        has_pose_3d = input_batch['has_pose_3d'].bool()
        has_disc_contact = input_batch['has_disc_contact'].bool()
        has_2d_keypoints_gtanno = input_batch['has_gt_kpts'].bool()
        has_smpl_ = input_batch['has_smpl'].bool() | input_batch['has_pgt_smpl'].bool()
        gt_keypoints_2d, gt_joints = input_batch['keypoints'], input_batch['pose_3d']
        gt_pose, gt_betas, gt_disc_contact = input_batch['pose'], input_batch['betas'], input_batch['contact_vec']
        gt_out = self.smpl(betas=gt_betas, body_pose=gt_pose[:, 3:], global_orient=gt_pose[:, :3])
        gt_model_joints, gt_verts = gt_out.joints, gt_out.vertices
        # keypoints from [-1, 1] to pixels (:131-134)
        gt_keypoints_2d_orig = gt_keypoints_2d.clone()
        gt_keypoints_2d_orig[:, :, :-1] = 0.5 * o.img_res * (gt_keypoints_2d_orig[:, :, :-1] + 1)

        # current best fits (:140-151)
        opt_pose, opt_betas = self.fits_dict[(dataset_name, indices, rot_angle, is_flipped)]
        opt_pose, opt_betas = opt_pose.to(dev), opt_betas.to(dev)
        opt_output = self.smpl(betas=opt_betas, body_pose=opt_pose[:, 3:], global_orient=opt_pose[:, :3])
        opt_vertices, opt_joints = opt_output.vertices, opt_output.joints
        opt_contact_l3 = self.contact_from_verts(opt_vertices, mode='regions')
        # camera translations by weighted least squares (:156-169), on the device
        gt_cam_t = estimate_translation(gt_model_joints, gt_keypoints_2d_orig, focal_length=self.focal_length,
                                        img_size=o.img_res, has_2d_kp_anno=has_2d_keypoints_gtanno)
        opt_cam_t = estimate_translation(opt_joints, gt_keypoints_2d_orig, focal_length=self.focal_length,
                                         img_size=o.img_res, has_2d_kp_anno=has_2d_keypoints_gtanno)
        centre = 0.5 * o.img_res * torch.ones(batch_size, 2, device=dev)
        opt_joint_loss = self.smplify.get_fitting_loss(opt_pose, opt_betas, opt_cam_t, centre, gt_keypoints_2d_orig,
                                                       has_2d_keypoints_gtanno).mean(dim=-1)

        def camera_from(pred_camera):                                        # :183-185, :213-216
            return torch.stack([pred_camera[:, 1], pred_camera[:, 2],
                                2 * self.focal_length / (o.img_res * pred_camera[:, 0] + 1e-9)], dim=-1)

        with torch.no_grad():                                                # the frozen SPIN model, for logging (:174-186)
            rot_spin, betas_spin, cam_spin = self.modelspin(images)
            spin_vertices = self.smpl(betas=betas_spin, body_pose=rot_spin[:, 1:], global_orient=rot_spin[:, 0].unsqueeze(1),
                                      pose2rot=False).vertices.clone()
            spin_cam_t = camera_from(cam_spin)

        # the regressor (:191-196)
        pred_rotmat, pred_betas, pred_camera = self.model(images)
        pred_output = self.smpl(betas=pred_betas, body_pose=pred_rotmat[:, 1:], global_orient=pred_rotmat[:, 0].unsqueeze(1),
                                pose2rot=False)
        pred_vertices, pred_joints = pred_output.vertices, pred_output.joints
        # rotation matrices -> axis-angle (:199-204), on the device
        pred_pose = rotation_matrix_to_angle_axis(pred_rotmat.detach().reshape(-1, 3, 3)).view(batch_size, -1)
        pred_pose = torch.nan_to_num(pred_pose, nan=0.0, posinf=float('inf'), neginf=float('-inf'))
        pred_cam_t = camera_from(pred_camera)
        pred_keypoints_2d = perspective_projection(pred_joints, torch.eye(3, device=dev).unsqueeze(0).expand(batch_size, -1, -1),
                                                   pred_cam_t, self.focal_length, camera_center)
        pred_keypoints_2d = pred_keypoints_2d / (o.img_res / 2.)

        smplifyoptiverts = None
        if o.run_smplify:                                                    # :226-265
            new_vertices, new_joints, new_pose, new_betas, new_cam_t, new_joint_loss, smplifyoptiverts = self.smplify(
                pred_pose.detach(), pred_betas.detach(), pred_cam_t.detach(), centre, gt_keypoints_2d_orig,
                use_contact=o.use_contact_in_the_loop, contactlist=self.contactlists, gt_contact=[gt_disc_contact, None],
                ignore_idxs=has_smpl_, has_discrete_contact=has_disc_contact, has_gt_keypoints=has_2d_keypoints_gtanno,
                contact_loss_weight=o.contact_in_the_loop_loss_weight, contact_loss_return='sum',
                segments=self.criterion_cospin.segments)
            new_joint_loss = new_joint_loss.mean(dim=-1)
            update = new_joint_loss <= opt_joint_loss
            new_contact_l3 = self.contact_from_verts(new_vertices, mode='regions')
            closer = ((gt_disc_contact * new_contact_l3) <= (gt_disc_contact * opt_contact_l3)).sum(1) > 0
            if o.use_contact_in_the_loop:
                update = torch.where(has_disc_contact, closer & update, update)
            pick = lambda new, old: torch.where(update.view(-1, *([1] * (old.dim() - 1))), new, old)
            opt_joint_loss, opt_vertices = pick(new_joint_loss, opt_joint_loss), pick(new_vertices, opt_vertices)
            opt_contact_l3, opt_joints = pick(new_contact_l3, opt_contact_l3), pick(new_joints, opt_joints)
            opt_pose, opt_betas, opt_cam_t = pick(new_pose, opt_pose), pick(new_betas, opt_betas), pick(new_cam_t, opt_cam_t)
            self.fits_dict[(dataset_name, indices, rot_angle, is_flipped, update)] = (opt_pose, opt_betas)

        # ground truth replaces the fit where it exists (:271-275)
        use_gt = lambda gt, opt: torch.where(has_smpl_.view(-1, *([1] * (opt.dim() - 1))), gt, opt)
        opt_cam_t, opt_joints = use_gt(gt_cam_t, opt_cam_t), use_gt(gt_model_joints, opt_joints)
        opt_pose, opt_betas, opt_vertices = use_gt(gt_pose, opt_pose), use_gt(gt_betas, opt_betas), use_gt(gt_verts, opt_vertices)
        valid_fit = (opt_joint_loss < o.smplify_threshold).to(dev)         # :278
        valid_fit_pose = has_smpl_ | valid_fit
        valid_fit_shape = has_smpl_ | valid_fit

        loss, loss_dict = self.criterion_cospin(pred_rotmat, pred_betas, opt_pose, opt_betas, pred_keypoints_2d,
                                                gt_keypoints_2d, pred_joints, gt_joints, has_pose_3d, pred_vertices,
                                                opt_vertices, pred_camera, valid_fit_pose, valid_fit_shape)
        losses = {'loss': loss.detach()}
        for k, val in loss_dict.items():
            losses[k] = val.detach()
        output = {'pred_vertices': pred_vertices.detach(), 'spin_vertices': spin_vertices,
                  'opt_vertices': opt_vertices.detach(), 'pred_cam_t': pred_cam_t.detach(), 'spin_cam_t': spin_cam_t,
                  'opt_cam_t': opt_cam_t.detach(), 'smplifyoptiverts': smplifyoptiverts, 'gt_contact_l3': gt_disc_contact,
                  'has_contact_pc': has_disc_contact, 'has_contact': has_disc_contact,
                  'valid_kpts_anno': valid_fit | has_smpl_, 'gt_keypoints': gt_keypoints_2d_orig}
        return loss, losses, output
