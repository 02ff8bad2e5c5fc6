"""The reference's stock-torch-op formulation of the contact forward, restated on CPU tensors --
TEST INFRASTRUCTURE / CPU BASELINE ONLY (never imported by tuch_amd/).

It materialises the same intermediates as the reference does -- three [1,V,V] matrices for the
pairwise distances (tuch/utils/contact.py:27-42) and the [1,Q,F,3,3] centred-triangle tensor
(contact.py:79, 3.4 GB at SMPL size) -- so that its timing is representative of
"the reference's PyTorch CPU path" (BASELINE.md §5).  Equivalence to the reference is covered
by tests/test_oracle_golden.py::test_torch_chain_matches_reference_goldens.
"""
from __future__ import annotations

import math

import torch


def pairwise_sq(x, y):
    """contact.py:23-47 with squared=True: rx^T + ry - 2 zz from three bmm's."""
    xx, yy, zz = x @ x.transpose(2, 1), y @ y.transpose(2, 1), x @ y.transpose(2, 1)
    rx = torch.diagonal(xx, dim1=1, dim2=2).unsqueeze(1).expand_as(zz.transpose(2, 1))
    ry = torch.diagonal(yy, dim1=1, dim2=2).unsqueeze(1).expand_as(zz)
    return rx.transpose(2, 1) + ry - 2 * zz


def winding(points, triangles):
    """contact.py:49-147: [B,Q,3], [B,F,3,3] -> [B,Q]."""
    centred = triangles[:, None] - points[:, :, None, None]
    norms = centred.norm(dim=-1)
    num = (centred[..., 0, :] * torch.cross(centred[..., 1, :], centred[..., 2, :], dim=-1)).sum(-1)
    d01 = (centred[..., 0, :] * centred[..., 1, :]).sum(-1)
    d12 = (centred[..., 1, :] * centred[..., 2, :]).sum(-1)
    d02 = (centred[..., 0, :] * centred[..., 2, :]).sum(-1)
    del centred
    den = norms.prod(dim=-1) + d01 * norms[..., 2] + d02 * norms[..., 1] + d12 * norms[..., 0]
    return (2 * torch.atan2(num, den)).sum(-1) / (4 * math.pi)


def contact_forward_one_body(verts, faces, geomask, euclthres):
    """One trip of the per-body loop of contact_fitting_loss (losses.py:74-105), segments off.
    verts [V,3] float32, faces [F,3] long, geomask [V,V] bool -> scalar contact term."""
    v = verts[None]
    dists = pairwise_sq(v, v)
    exterior = winding(v, verts[faces][None]).squeeze() <= 0.99
    dists[:, ~geomask] = float('inf')
    arg = torch.argmin(dists, dim=1)[0]
    d = torch.norm(verts - verts[arg], dim=1)
    inside = (torch.tanh(d[~exterior] / 0.04) ** 2).sum()
    sel = exterior & (d < euclthres)
    outside = (0.005 * torch.tanh(d[sel] / 0.005) ** 2).sum()
    return inside + outside
