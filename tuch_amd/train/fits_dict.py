"""Drop-in for the reference's ``tuch/train/fits_dict.py``: the dictionary of the best SMPL fit per training image
(SURVEY.md §8f-2), kept ON THE DEVICE.

The reference holds one CPU tensor per dataset and, on every training step, gathers / scatters the batch's rows in
a Python loop (fits_dict.py:59-85), rotates the global orientation through torchgeometry + a per-sample
``cv2.Rodrigues`` on the host (:97-119) and moves the result to the GPU.  Here the tables live on the device, the
gather / scatter are single indexed copies, and the rotation is composed and converted back to axis-angle by the HIP
kernel behind ``rotation_matrix_to_angle_axis`` -- no host round trip.  Same constructor, ``__getitem__`` /
``__setitem__`` keys, ``save``, ``flip_pose`` and ``rotate_pose`` as the reference.

cv2.Rodrigues(R) and torchgeometry's rotation_matrix_to_angle_axis both return the rotation vector of R; they differ
only in how they round at angles within ~1e-3 of 0 or pi.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from ..utils.geometry import angle_axis_to_rotation_matrix, rotation_matrix_to_angle_axis

# SPIN's constants.SMPL_POSE_FLIP_PERM (left/right swapped joints, three axis-angle entries each)
SMPL_JOINTS_FLIP_PERM = [0, 2, 1, 3, 5, 4, 6, 8, 7, 9, 11, 10, 12, 14, 13, 15, 17, 16, 19, 18, 21, 20, 23, 22]
SMPL_POSE_FLIP_PERM = [3 * i + k for i in SMPL_JOINTS_FLIP_PERM for k in range(3)]


def _flip_perm():
    from ..models.smpl import reference_constants
    c = reference_constants()
    return list(getattr(c, 'SMPL_POSE_FLIP_PERM', SMPL_POSE_FLIP_PERM)) if c is not None else SMPL_POSE_FLIP_PERM


class FitsDict():
    """Dictionary keeping track of the best fit per image in the training set (fits_dict.py:29-119)."""

    def __init__(self, options, train_dataset, device=None):
        from ..assets import config_path
        self.options = options
        self.train_dataset = train_dataset
        self.device = torch.device(device) if device is not None else torch.device('cuda')
        self.fits_dict = {}
        self.flipped_parts = torch.tensor(_flip_perm(), dtype=torch.int64, device=self.device)
        self._ds_index = {}
        self._group_cache = {}
        self._winner = {}             # per dataset: scratch column of __setitem__ (all -1 between calls)
        for ds_name, ds in train_dataset.dataset_dict.items():             # fits_dict.py:38-52
            dict_file = os.path.join(options.checkpoint_dir, ds_name + '_fits.npy')
            if not os.path.isfile(dict_file):
                dict_file = os.path.join(config_path('STATIC_FITS_DIR'), ds_name + '_fits.npy')
            if os.path.isfile(dict_file):
                table = torch.from_numpy(np.load(dict_file)).to(torch.float32)
            else:                                                           # no static fit: mean pose
                table = torch.zeros((len(train_dataset.datasets[train_dataset.dataset_dict[ds_name]]), 82))
            self._ds_index[ds_name] = len(self._ds_index)
            self.fits_dict[ds_name] = table.to(self.device)

    def save(self):
        """Save dictionary state to disk (fits_dict.py:54-58)."""
        for ds_name in self.train_dataset.dataset_dict.keys():
            np.save(os.path.join(self.options.checkpoint_dir, ds_name + '_fits.npy'), self.fits_dict[ds_name].cpu().numpy())

    def _groups(self, dataset_name, ind):
        """(dataset, positions in the batch, rows of its table) per dataset present in the batch."""
        ind = torch.as_tensor(ind).to(self.device, torch.int64)
        names = tuple(dataset_name)
        groups = self._group_cache.get(names)
        if groups is None:            # batch positions per dataset: uploaded once per distinct composition of a batch
            groups = [(ds, torch.tensor([n for n, d in enumerate(names) if d == ds], dtype=torch.int64, device=self.device))
                      for ds in dict.fromkeys(names)]
            if len(self._group_cache) >= 64:
                self._group_cache.pop(next(iter(self._group_cache)))
            self._group_cache[names] = groups
        for ds, pos in groups:
            yield ds, pos, ind[pos]

    def __getitem__(self, x):
        """(pose [B,72], betas [B,10]) of the batch's images, flipped / rotated like the images (fits_dict.py:59-73)."""
        dataset_name, ind, rot, is_flipped = x
        params = torch.zeros((len(dataset_name), 82), dtype=torch.float32, device=self.device)
        for ds, pos, rows in self._groups(dataset_name, ind):
            params[pos] = self.fits_dict[ds][rows]
        rot = torch.as_tensor(rot).to(self.device, torch.float32)
        pose = self.flip_pose(self.rotate_pose(params[:, :72], rot), torch.as_tensor(is_flipped).to(self.device))
        return pose, params[:, 72:].clone()

    def __setitem__(self, x, val):
        """Write back the rows selected by ``update`` after undoing flip and rotation (fits_dict.py:75-85)."""
        dataset_name, ind, rot, is_flipped, update = x
        pose, betas = val
        pose, betas = pose.to(self.device, torch.float32), betas.to(self.device, torch.float32)
        rot = torch.as_tensor(rot).to(self.device, torch.float32)
        update = torch.as_tensor(update).to(self.device).bool()
        pose = self.rotate_pose(self.flip_pose(pose, torch.as_tensor(is_flipped).to(self.device)), -rot)
        params = torch.cat((pose, betas), dim=-1)
        for ds, pos, rows in self._groups(dataset_name, ind):
            table = self.fits_dict[ds]
            # The reference writes sample by sample, in batch order, only where ``update`` is set: if a row occurs twice
            # in the batch (MixedDataset wraps small datasets) the LAST updating sample wins and a non-updating one
            # never writes.  Same result without a host sync and without a write race: per table row the largest batch
            # position that updates it (scatter-max into a persistent scratch column, -1 = nobody), then EVERY
            # occurrence of the row writes that winner's value (or the old value) -- duplicates write identical data.
            winner = self._winner.get(ds)
            if winner is None:
                winner = self._winner[ds] = torch.full((table.shape[0],), -1, dtype=torch.int64, device=self.device)
            local = torch.arange(pos.shape[0], dtype=torch.int64, device=self.device)
            winner.scatter_reduce_(0, rows, torch.where(update[pos], local, torch.full_like(local, -1)), 'amax')
            w = winner[rows]
            table[rows] = torch.where((w >= 0)[:, None], params[pos][w.clamp(min=0)], table[rows])
            winner.index_fill_(0, rows, -1)          # (not `winner[rows] = -1`: that uploads a host scalar -- no hipGraph capture)

    def flip_pose(self, pose, is_flipped):
        """Flip SMPL pose parameters: swap left / right joints, negate the 2nd and 3rd axis-angle entries (:87-95)."""
        flipped = pose[:, self.flipped_parts.to(pose.device)]
        sign = torch.ones(72, dtype=pose.dtype, device=pose.device)
        sign[1::3] = -1
        sign[2::3] = -1
        return torch.where(is_flipped.bool()[:, None], flipped * sign, pose)

    def rotate_pose(self, pose, rot):
        """Rotate the global orientation by ``rot`` degrees about the camera axis (fits_dict.py:97-119)."""
        pose = pose.clone()
        ang = -np.pi * rot / 180.
        cos, sin = torch.cos(ang), torch.sin(ang)
        zeros, ones = torch.zeros_like(cos), torch.ones_like(cos)
        R = torch.stack([cos, -sin, zeros, sin, cos, zeros, zeros, zeros, ones], dim=-1).view(-1, 3, 3)
        composed = torch.matmul(R, angle_axis_to_rotation_matrix(pose[:, :3])[:, :3, :3])
        pose[:, :3] = rotation_matrix_to_angle_axis(composed.contiguous()).to(pose.dtype)
        return pose
