"""Drop-in for the reference's ``tuch/eft/loss.py``: the EFT fitting loss with the self-contact
term on the HIP kernels.  Same constructor and method signatures; the DSC region tables and the
segments may be injected (``cdict=``, ``segments=``) because the asset files do not ship.

Contact term (reference eft/loss.py:129-181), per body:
    contact = mean_{interior} tanh^2(d/0.04) + mean_{exterior} 0.005 tanh^2(d/0.005)
    r2r     = sum over annotated region pairs of the masked minimum squared distance
    total   = sum_b 100 * (contact + 0.5 * r2r)
i.e. the SMPLify terms with means instead of sums and no distance gate on the exterior term.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..utils.geometry import perspective_projection


class EFTLoss(nn.Module):
    def __init__(self, options, device, smpl, num_verts, faces, geodistssmpl, geothres, face_tensor=None,
                 use_hd=True, keypoint_weight=1.0, shape_weight=1.0, contact_weight=1.0,
                 cdict=None, segments=None, dsc_root=None):
        super().__init__()
        self.device = device
        self.options = options
        self.focal_length = 5000
        self.camera_center = torch.zeros(2, device=device)
        self.criterion_shape = nn.L1Loss().to(self.device)
        self.criterion_keypoints = nn.MSELoss(reduction='none').to(self.device)
        self.keypoints_weight = keypoint_weight
        self.shape_weight = shape_weight
        self.contact_weight = contact_weight
        self.face_tensor = face_tensor
        self.geodistssmpl = geodistssmpl
        self.geothres = geothres
        self.geomask = self.geodistssmpl > self.geothres
        if cdict is None:                                          # eft/loss.py:63-66
            from ..assets import load_contact_regions
            cdict = load_contact_regions(dsc_root)
        self.cdict = cdict
        ft = face_tensor[0] if face_tensor.dim() == 3 else face_tensor
        if segments is None:                                       # eft/loss.py:69-71: always built
            from ..utils.segmentation import BatchBodySegment, reference_segment_names
            segments = BatchBodySegment(reference_segment_names(), ft)
        self.segments = segments
        regions, pairs = ops.region_tables(cdict)
        self._model = ops.ContactModel(ft, self.geomask, segments.tables(), regions, pairs, device=ft.device)

    def forward(self, body, camera, batch):
        """Reference: eft/loss.py:73-118 (the debugging print is not reproduced)."""
        batch_size = camera.shape[0]
        gt_keypoints, gt_contact = batch['keypoints'], batch['contact']
        rotation = torch.eye(3, device=self.device).unsqueeze(0).expand(batch_size, -1, -1)
        res = self.options.img_res
        camera_t = torch.stack([camera[:, 1], camera[:, 2],
                                2 * self.focal_length / (res * camera[:, 0] + 1e-9)], dim=-1)
        pred = perspective_projection(body.joints, rotation, camera_t, self.focal_length,
                                      self.camera_center[None].expand(batch_size, -1))
        pred = 0.5 * res * (pred / (res / 2.) + 1)
        gt = gt_keypoints.clone()
        gt[:, :, :-1] = 0.5 * res * (gt[:, :, :-1] + 1)
        loss_keypoints = self.keypoint_loss(pred, gt) * self.keypoints_weight
        loss_shape = torch.mean(body.betas ** 2) * self.shape_weight
        loss_contact = torch.tensor(0.0, device=self.device)
        if self.contact_weight > 0:
            loss_contact = self.contact_loss(gt_contact, body.vertices) * self.contact_weight
        loss = 60 * (loss_keypoints + loss_shape + loss_contact)
        return loss, {'loss_shape': loss_shape, 'loss_keypoints': loss_keypoints, 'loss_contact': loss_contact}

    def keypoint_loss(self, pred_keypoints_2d, gt_keypoints_2d):
        conf = gt_keypoints_2d[:, :, -1].unsqueeze(-1).clone()
        return (conf * self.criterion_keypoints(pred_keypoints_2d, gt_keypoints_2d[:, :, :-1])).mean(axis=(1, 2)).mean()

    def contact_loss(self, gt_contact, verts):
        model = self._model
        exterior, _, partner, _ = model.exterior_and_partner(verts, apply_segments=True)
        _, terms = ops.contact_terms(verts, partner, exterior, None, ops.MODE_TRAIN, 0.0)
        n_ext = exterior.to(torch.float32).sum(dim=1)
        n_int = exterior.shape[1] - n_ext
        contact = terms[:, 0] / n_int.clamp(min=1.0) + terms[:, 1] / n_ext.clamp(min=1.0)
        r2r, _ = model.region_pair_min(verts, select=(gt_contact == 1), masked=True)
        return (100 * (contact + 0.5 * r2r.sum(dim=1))).sum()
