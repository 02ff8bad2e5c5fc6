"""Drop-in for the reference's ``tuch/utils/contact.py`` (same names, argument meaning
and return shapes), running on the HIP kernels of libtuch_amd.so.

The three functions below keep the reference's *materialising* signatures for callers
that want the full matrices.  The loss code in this package does not use them: it calls
the fused kernels (ops.ContactModel) that never build anything of size VxV or QxF.
"""
from __future__ import annotations

import torch

from .. import ops


def batch_pairwise_dist(x, y, use_cuda=True, squared=True):
    """Reference: tuch/utils/contact.py:23-47.  [B,Nx,3],[B,Ny,3] -> [B,Nx,Ny].
    ``use_cuda`` is accepted for signature compatibility; the device is the inputs'."""
    return ops.batch_pairwise_dist(x, y, squared=squared)


def solid_angles(points, triangles, thresh: float = 1e-8):
    """Reference: tuch/utils/contact.py:49-109.  [B,Q,3],[B,F,3,3] -> [B,Q,F]
    (``thresh`` is unused there as well)."""
    return ops.solid_angles(points, triangles)


def winding_numbers(points, triangles, thresh: float = 1e-8):
    """Reference: tuch/utils/contact.py:112-147.  [B,Q,3],[B,F,3,3] -> [B,Q]."""
    return ops.winding_numbers(points, triangles)
