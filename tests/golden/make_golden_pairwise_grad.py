#!/usr/bin/env python3
"""Golden gradients through batch_pairwise_dist, produced by the REFERENCE's own function
(/root/reference/tuch/utils/contact.py:23-47, imported, never copied) and torch autograd on the CPU:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_pairwise_grad.py

Writes tests/golden/pairwise_grad.npz (inputs next to expected outputs).  Three uses:
  * two different point sets, a random cotangent, squared and not squared (the general adjoint);
  * one tensor passed as both arguments (tuch/smplify/losses.py:76-78), loss = sum of block minima as the region-to-region
    term takes them (losses.py:112-116);
  * a region-sized batch as contact_from_verts calls it (tuch/train/train_module.py:83-88).
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(1, '/root/reference')

import numpy as np
import torch

torch.cuda.LongTensor = torch.LongTensor            # the reference asks for a CUDA index type by default
from tuch.utils import contact as ref               # noqa: E402

rng = np.random.default_rng(2024)
out = {}

# general adjoint
B, NX, NY = 3, 70, 130
x = rng.standard_normal((B, NX, 3)).astype(np.float32)
y = (rng.standard_normal((B, NY, 3)) + 0.25).astype(np.float32)
G = rng.standard_normal((B, NX, NY)).astype(np.float32)
out.update(x=x, y=y, G=G)
for squared in (True, False):
    xt, yt = torch.tensor(x, requires_grad=True), torch.tensor(y, requires_grad=True)
    P = ref.batch_pairwise_dist(xt, yt, use_cuda=False, squared=squared)
    (P * torch.tensor(G)).sum().backward()
    tag = 'sq' if squared else 'root'
    out['P_' + tag] = P.detach().numpy()
    out['gx_' + tag] = xt.grad.numpy()
    out['gy_' + tag] = yt.grad.numpy()

# one tensor as both arguments, block minima (the r2r term)
V = 300
v = (rng.standard_normal((1, V, 3)) * 0.3).astype(np.float32)
blocks = [(rng.choice(V, 20, replace=False), rng.choice(V, 25, replace=False)) for _ in range(4)]
vt = torch.tensor(v, requires_grad=True)
P = ref.batch_pairwise_dist(vt[[0]], vt[[0]], use_cuda=False, squared=True)
loss = 0
for r1, r2 in blocks:
    loss = loss + torch.min(P[:, r1, :][:, :, r2])
loss.backward()
out.update(v=v, r2r_loss=np.float32(loss.item()), r2r_grad=vt.grad.numpy(),
           blocks_a=np.stack([b[0] for b in blocks]), blocks_b=np.stack([b[1] for b in blocks]))

# a wide, short batch (many bodies, one region pair)
Bw = 17
xa = rng.standard_normal((Bw, 9, 3)).astype(np.float32)
ya = rng.standard_normal((Bw, 257, 3)).astype(np.float32)
xat, yat = torch.tensor(xa, requires_grad=True), torch.tensor(ya, requires_grad=True)
d = ref.batch_pairwise_dist(xat, yat, use_cuda=False, squared=True)
d.reshape(Bw, -1).min(1)[0].sum().backward()
out.update(xa=xa, ya=ya, gxa=xat.grad.numpy(), gya=yat.grad.numpy())

np.savez_compressed(os.path.join(HERE, 'pairwise_grad.npz'), **out)
print('wrote pairwise_grad.npz', {k: v.shape for k, v in out.items() if hasattr(v, 'shape')})
