"""Host simulation 2: ray direction chosen per 64-query block among the 6 axis directions of a generic rotated
frame (the one closest to the block's mean outward normal); 9-slab ray/volume tests."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from synthetic import make_body, random_poses
from tuch_amd import ops
from oracle import lbs as ol
body = make_body(84, 82)
leaf_faces = int(sys.argv[1]) if len(sys.argv) > 1 else 64
qsize = int(sys.argv[2]) if len(sys.argv) > 2 else 64
t = ops.cluster_tree(body.faces, body.num_verts, leaf_faces)
nodes, qperm, face_leaf = t['nodes'], t['qperm'][:body.num_verts], t['face_leaf']
leaf_ids = [i for i in range(len(nodes)) if nodes[i, 3] > 0]
leaf_len = np.array([nodes[i, 3] for i in leaf_ids])
nleaf = len(leaf_ids)
B = 8
bp, go, be = random_poses(B, 1002)
m = ol.model_tensors(body)
verts, _ = ol.smpl_forward(m, torch.tensor(be), torch.tensor(bp), torch.tensor(go))
verts = verts.numpy().astype(np.float64)
# generic rotation
rng = np.random.default_rng(5)
Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
C = np.array([[1,0,0],[0,1,0],[0,0,1],[1,1,0],[1,-1,0],[1,0,1],[1,0,-1],[0,1,1],[0,1,-1]], float)   # 9 slab functionals
tot_elems = tot_blocks = 0
near_counts = []
for b in range(B):
    v = verts[b] @ Q.T
    fv = v[body.faces]
    fn = np.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0])
    vn = np.zeros_like(v)
    for k in range(3):
        np.add.at(vn, body.faces[:, k], fn)
    fp = fv @ C.T                                    # [F,3,9]
    lo = np.full((nleaf, 9), np.inf); hi = np.full((nleaf, 9), -np.inf)
    np.minimum.at(lo, face_leaf, fp.min(1)); np.maximum.at(hi, face_leaf, fp.max(1))
    q = v[qperm]; qn = vn[qperm]
    qp = q @ C.T
    for blk in range(0, len(q), qsize):
        nrm = qn[blk:blk + qsize].sum(0)
        a = int(np.argmax(np.abs(nrm))); s = np.sign(nrm[a])
        rate = s * C[:, a]                            # d f_k / dt along the ray
        qq = qp[blk:blk + qsize]                      # [64,9]
        ok_lo = (qq[:, None, :] >= lo[None]) | (rate[None, None, :] > 0)     # lower bound irrelevant if f increases
        ok_hi = (qq[:, None, :] <= hi[None]) | (rate[None, None, :] < 0)
        near = (ok_lo & ok_hi).all(2).any(0)
        near_counts.append(int(near.sum()))
        tot_elems += int(leaf_len[near].sum())
        tot_blocks += 1
print('leaf_faces %d (%d leaves), %d-query blocks: near leaves per block mean %.1f (p50 %d p90 %d max %d), leaf elements per block %.0f -> per body %.3f M element steps'
      % (leaf_faces, nleaf, qsize, np.mean(near_counts), np.percentile(near_counts, 50), np.percentile(near_counts, 90), max(near_counts),
         tot_elems / tot_blocks, tot_elems / B / 1e6))
