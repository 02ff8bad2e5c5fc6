"""Host simulation behind the 2D-grid form of the ray-crossing inside test (round 6): with all rays parallel, a
triangle can only be crossed by queries inside its projection -- bin the projected triangles of a posed body into a
uniform grid of the sheared (x', y') plane and let every query test the triangles of its own cell.
Prints, per grid resolution: entries per triangle, entries per body, list length per query (mean / p95 / max) and the
mean over wavefronts (64 consecutive queries in tree order) of the LONGEST list in the wavefront = the trips a
lane-per-query kernel makes.      python tools/diag/ray_grid_sim.py [bodies]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from synthetic import make_body, random_poses
from tuch_amd import ops
from oracle import lbs as olbs

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
body = make_body()
m = olbs.model_tensors(body)
bp, go, be = random_poses(64, 1002)
v, _ = olbs.smpl_forward(m, torch.tensor(be[:nb]), torch.tensor(bp[:nb]), torch.tensor(go[:nb]))
verts = v.numpy().astype(np.float32)
faces = body.faces.astype(np.int64)
tree = ops.cluster_tree(faces, body.num_verts, 32)
qperm = tree['qperm'][:body.num_verts]
KX, KY = 0.3217, 0.4331
for G in (64, 96, 128, 192):
    tot_e, per_q, wave_max = [], [], []
    for b in range(nb):
        x = verts[b, :, 0] - KX * verts[b, :, 2]
        y = verts[b, :, 1] - KY * verts[b, :, 2]
        lo = np.array([x.min(), y.min()]); ext = max(x.max() - x.min(), y.max() - y.min()) * 1.0001
        c = ext / G
        cx = np.floor((x - lo[0]) / c).astype(int); cy = np.floor((y - lo[1]) / c).astype(int)
        fx0 = cx[faces].min(1); fx1 = cx[faces].max(1); fy0 = cy[faces].min(1); fy1 = cy[faces].max(1)
        n_cells = (fx1 - fx0 + 1) * (fy1 - fy0 + 1)
        tot_e.append(n_cells.sum())
        cnt = np.zeros((G + 1, G + 1), np.int64)
        for f in range(len(faces)):
            cnt[fx0[f]:fx1[f] + 1, fy0[f]:fy1[f] + 1] += 1
        lq = cnt[cx, cy]
        per_q.append(lq)
        lt = lq[qperm]
        pad = (-len(lt)) % 64
        lt = np.concatenate([lt, np.zeros(pad, np.int64)]).reshape(-1, 64)
        wave_max.append(lt.max(1))
    per_q = np.concatenate(per_q); wave_max = np.concatenate(wave_max)
    print('G=%3d cell %.1f mm: entries/tri %.2f  entries/body %d  list per query mean %.1f p95 %d max %d   wave max: mean %.1f p95 %d max %d'
          % (G, c * 1e3, np.mean(tot_e) / len(faces), np.mean(tot_e), per_q.mean(), np.percentile(per_q, 95), per_q.max(),
             wave_max.mean(), np.percentile(wave_max, 95), wave_max.max()))
