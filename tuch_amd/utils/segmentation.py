"""Drop-in for the reference's ``tuch/utils/segmentation.py``: closed per-body-segment
sub-meshes that whitelist self-intersection inside one segment (skin folds at joints).

The reference reads each segment's vertex set from a painted .ply through trimesh
(segmentation.py:40-42) and its boundary loops from data.essentials.segments.smpl.segm_utils
(:45-46).  The reference's constructor calls work verbatim --

    BatchBodySegment([x for x in exn.segments.keys()], face_tensor[0])      # demo_smplify_dc.py:87, loss.py:91

-- and load exactly those files (config.SEGMENT_DIR; the .ply is parsed here, no trimesh needed).
Neither asset ships, so the same information may also be passed as plain arrays
(``segment_vidx=`` / ``bands=``, ``segments=``).  Missing assets raise; nothing is skipped silently.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from .. import ops


def reference_segment_names() -> List[str]:
    """``[x for x in exn.segments.keys()]`` (loss.py:91, demo_smplify_dc.py:87)."""
    from ..assets import reference_segm_utils
    return list(reference_segm_utils().segments.keys())


class BodySegment(nn.Module):
    """One named segment.  Attributes match the reference: ``name``, ``segment_vidx``,
    ``bands``, ``bands_verts``, ``bands_faces``, ``segment_faces``, ``append_idx``."""

    def __init__(self, name, faces, append_idx=None, segment_vidx=None, bands: Optional[Dict] = None):
        super().__init__()
        faces = faces.squeeze()
        self.device = faces.device
        self.name = name
        self.append_idx = int(faces.max().item()) if append_idx is None else append_idx
        if segment_vidx is None:               # segmentation.py:40-42
            from ..assets import config_path, read_ply_vertex_red
            red = read_ply_vertex_red(os.path.join(config_path('SEGMENT_DIR'), 'smpl_segment_{}.ply'.format(name)))
            segment_vidx = np.where(red == 255)[0]
        if bands is None:                      # segmentation.py:45-46
            from ..assets import reference_segm_utils
            bands = reference_segm_utils().segments[name]
        self.segment_vidx = np.asarray(segment_vidx, dtype=np.int64)
        self.bands = list(bands.keys())
        self.bands_verts = [np.asarray(v, dtype=np.int64) for v in bands.values()]
        closed = ops.segment_faces(faces.cpu().numpy(), self.segment_vidx, self.bands_verts,
                                   self.append_idx + 1)
        n_caps = sum(len(b) - 1 for b in self.bands_verts)
        self.register_buffer('bands_faces', torch.as_tensor(closed[len(closed) - n_caps:], device=self.device))
        self.register_buffer('segment_faces', torch.as_tensor(closed, device=self.device))

    @classmethod
    def from_reference_assets(cls, name, faces, segment_dir, segm_utils_segments, append_idx=None):
        from ..assets import read_ply_vertex_red      # no trimesh needed: only the red channel is used
        red = read_ply_vertex_red(os.path.join(segment_dir, 'smpl_segment_{}.ply'.format(name)))
        return cls(name, faces, append_idx, np.where(red == 255)[0], segm_utils_segments[name])

    def get_closed_segment(self, vertices):
        """[B,V,3] -> triangles [B,Fs,3,3] of the closed segment (segmentation.py:68-79)."""
        v = vertices.detach()
        caps = [v[:, torch.as_tensor(b, device=v.device)].mean(1, keepdim=True) for b in self.bands_verts]
        ext = torch.cat([v] + caps, 1)
        return ops.gather_triangles(ext, self.segment_faces.to(torch.int32))

    def has_self_isect(self, vertices):
        """exterior flags of the segment's own vertices w.r.t. its closed mesh (:81-99)."""
        v = vertices.detach()
        pts = v[:, torch.as_tensor(self.segment_vidx, device=v.device)].contiguous()
        return (ops.winding_numbers(pts, self.get_closed_segment(v)).squeeze() <= 0.99)


class BatchBodySegment(nn.Module):
    """Reference: segmentation.py:102-124.  ``segments`` maps name -> {'vidx', 'bands'}."""

    def __init__(self, names, faces, segments: Optional[Dict[str, dict]] = None):
        super().__init__()
        self.names = list(names)
        self.nv = int(faces.max().item())
        self.segmentation = {}
        for name in self.names:
            if segments is None:               # the reference's call: everything from the asset files
                self.segmentation[name] = BodySegment(name, faces)
            else:
                self.segmentation[name] = BodySegment(name, faces, None, segments[name]['vidx'],
                                                      segments[name]['bands'])

    def tables(self):
        """(vidx, [band loops]) per segment, the form ops.ContactModel consumes."""
        return [(s.segment_vidx, s.bands_verts) for s in self.segmentation.values()]

    def batch_has_self_isec(self, vertices):
        return [seg.has_self_isect(vertices) for seg in self.segmentation.values()]
