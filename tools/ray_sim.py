"""Host simulation: how many leaf-strip elements would a +z ray-crossing walk visit per 64-query block
(vs the solid-angle tree walk's leaf + cap elements)?  Uses the model's cluster tree and oracle-posed bodies."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from synthetic import make_body, random_poses
from tuch_amd import ops
from oracle import lbs as ol
body = make_body(84, 82)
t = ops.cluster_tree(body.faces, body.num_verts)
nodes, qperm, face_leaf = t['nodes'], t['qperm'][:body.num_verts], t['face_leaf']
leaf_ids = [i for i in range(len(nodes)) if nodes[i, 3] > 0]
leaf_len = {seq: nodes[i, 3] for seq, i in enumerate(leaf_ids)}
B = 8
bp, go, be = random_poses(B, 1002)
m = ol.model_tensors(body)
verts, _ = ol.smpl_forward(m, torch.tensor(be), torch.tensor(bp), torch.tensor(go))
verts = verts.numpy()
nleaf = face_leaf.max() + 1
for name, k in (('+z', (0.0, 0.0)), ('sheared', (0.3217, 0.4331))):
    tot_elems, tot_blocks, near_counts = 0, 0, []
    for b in range(B):
        v = verts[b].copy()
        v[:, 0] -= k[0] * v[:, 2]; v[:, 1] -= k[1] * v[:, 2]
        fv = v[body.faces]                       # [F,3,3]
        proj = lambda p: np.stack([p[..., 0], p[..., 1], p[..., 0] + p[..., 1], p[..., 0] - p[..., 1]], -1)
        fp = proj(fv)                            # [F,3,4]
        lo = np.full((nleaf, 4), np.inf); hi = np.full((nleaf, 4), -np.inf); zhi = np.full(nleaf, -np.inf)
        np.minimum.at(lo, face_leaf, fp.min(1)); np.maximum.at(hi, face_leaf, fp.max(1))
        np.maximum.at(zhi, face_leaf, fv[..., 2].max(1))
        # extra ray slabs: x+z, x-z, y+z, y-z
        xz = np.full((nleaf, 4), 0.0)
        a = fv[..., 0] + fv[..., 2]; bq = fv[..., 0] - fv[..., 2]; c = fv[..., 1] + fv[..., 2]; d = fv[..., 1] - fv[..., 2]
        hi_xpz = np.full(nleaf, -np.inf); lo_xmz = np.full(nleaf, np.inf); hi_ypz = np.full(nleaf, -np.inf); lo_ymz = np.full(nleaf, np.inf)
        np.maximum.at(hi_xpz, face_leaf, a.max(1)); np.minimum.at(lo_xmz, face_leaf, bq.min(1))
        np.maximum.at(hi_ypz, face_leaf, c.max(1)); np.minimum.at(lo_ymz, face_leaf, d.min(1))
        q = v[qperm]
        qp = proj(q)
        for blk in range(0, len(q), 64):
            qq, qz, q3 = qp[blk:blk + 64], q[blk:blk + 64, 2], q[blk:blk + 64]
            inside = ((qq[:, None, :] >= lo[None]) & (qq[:, None, :] <= hi[None])).all(2) & (qz[:, None] <= zhi[None])
            inside &= (q3[:, 0] + q3[:, 2])[:, None] <= hi_xpz[None]
            inside &= (q3[:, 0] - q3[:, 2])[:, None] >= lo_xmz[None]
            inside &= (q3[:, 1] + q3[:, 2])[:, None] <= hi_ypz[None]
            inside &= (q3[:, 1] - q3[:, 2])[:, None] >= lo_ymz[None]
            near = inside.any(0)
            near_counts.append(int(near.sum()))
            tot_elems += sum(leaf_len[s] for s in np.where(near)[0])
            tot_blocks += 1
    print('%s: blocks %d, near leaves per block mean %.1f (p90 %d, max %d), leaf elements per block %.0f  -> per body %.2f M element steps (tree walk today: ~0.27 M per body)'
          % (name, tot_blocks, np.mean(near_counts), np.percentile(near_counts, 90), max(near_counts), tot_elems / tot_blocks,
             tot_elems / B / 1e6))
