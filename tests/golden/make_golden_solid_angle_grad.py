#!/usr/bin/env python3
"""Golden gradients through solid_angles / winding_numbers, produced by the REFERENCE's own functions
(/root/reference/tuch/utils/contact.py:49-147, imported, never copied) and torch autograd on the CPU:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_solid_angle_grad.py

Writes tests/golden/solid_angle_grad.npz (inputs next to expected outputs).  The reference itself only calls these two under
torch.no_grad(); they are plain differentiable torch ops all the same, and a user of tuch.utils.contact may differentiate.
Query points OFF the triangles (on a triangle's corner the reference's own gradient is NaN: atan2 at (0, 0))."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(1, '/root/reference')

import numpy as np
import torch

from tuch.utils import contact as ref               # noqa: E402

rng = np.random.default_rng(4711)
B, Q, F = 2, 150, 97
pts = rng.standard_normal((B, Q, 3)).astype(np.float32)
tris = (rng.standard_normal((B, F, 3, 3)) * 0.8 + rng.standard_normal((B, F, 1, 3))).astype(np.float32)
G = rng.standard_normal((B, Q, F)).astype(np.float32)
gw = rng.standard_normal((B, Q)).astype(np.float32)
out = dict(points=pts, triangles=tris, G=G, gw=gw)

p, t = torch.tensor(pts, requires_grad=True), torch.tensor(tris, requires_grad=True)
sa = ref.solid_angles(p, t)
(sa * torch.tensor(G)).sum().backward()
out.update(solid_angles=sa.detach().numpy(), sa_grad_points=p.grad.numpy(), sa_grad_triangles=t.grad.numpy())

p, t = torch.tensor(pts, requires_grad=True), torch.tensor(tris, requires_grad=True)
w = ref.winding_numbers(p, t)
(w * torch.tensor(gw)).sum().backward()
out.update(winding=w.detach().numpy(), w_grad_points=p.grad.numpy(), w_grad_triangles=t.grad.numpy())

# a closed mesh (octahedron) with query points inside and outside, and only the points differentiated
verts = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32) * 0.7
faces = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]])
octa = verts[faces][None]
q2 = (rng.standard_normal((1, 40, 3)) * 0.6).astype(np.float32)
p2 = torch.tensor(q2, requires_grad=True)
w2 = ref.winding_numbers(p2, torch.tensor(octa))
(w2 ** 2).sum().backward()
out.update(octa=octa, octa_points=q2, octa_winding=w2.detach().numpy(), octa_grad_points=p2.grad.numpy())

np.savez_compressed(os.path.join(HERE, 'solid_angle_grad.npz'), **out)
print('wrote solid_angle_grad.npz', {k: v.shape for k, v in out.items()}, float(np.abs(out['sa_grad_points']).max()))
