"""Host simulation: what would splitting every leaf's strip run into k sub-runs with slabs of their own save the
crossing walk?  Counts wave-elements (64 rays x one strip element) of the leaf-major walk: per leaf ceil(rays / 64) x len,
against per sub-run ceil(rays passing the sub-run's slabs / 64) x (len_k + 2 priming elements).
    python tools/diag/ray_subrun_sim.py [bodies]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from synthetic import make_body, random_poses
from tuch_amd import ops
from oracle import lbs as olbs

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
body = make_body()
m = olbs.model_tensors(body)
bp, go, be = random_poses(64, 1002)
v, _ = olbs.smpl_forward(m, torch.tensor(be[:nb]), torch.tensor(bp[:nb]), torch.tensor(go[:nb]))
verts = v.numpy().astype(np.float64)
tree = ops.cluster_tree(body.faces, body.num_verts, 32)
nodes, vidx = tree['nodes'], tree['vidx']
leaves = [n for n in range(len(nodes)) if nodes[n, 5] < 0 and nodes[n, 3] > 0]
print('leaves', len(leaves), 'mean strip len', np.mean([nodes[n, 3] for n in leaves]))
KX, KY = 0.3217, 0.4331


def proj(p):
    x, y, z = p[:, 0] - KX * p[:, 2], p[:, 1] - KY * p[:, 2], p[:, 2]
    return np.stack([x, y, z, x + y, x - y, x + z, x - z, y + z, y - z], 1)


def passing(pq, pe):
    lo, hi = pe.min(0), pe.max(0)
    ok = np.ones(len(pq), bool)
    for k in (0, 1, 3, 4):
        ok &= (pq[:, k] >= lo[k]) & (pq[:, k] <= hi[k])
    for k in (2, 5, 7):
        ok &= pq[:, k] <= hi[k]
    for k in (6, 8):
        ok &= pq[:, k] >= lo[k]
    return ok


res = {}
for b in range(nb):
    pq = proj(verts[b])
    for n in leaves:
        off, ln = nodes[n, 2], nodes[n, 3]
        pe = proj(verts[b][vidx[off:off + ln]])
        ok = passing(pq, pe)
        r = int(ok.sum())
        res.setdefault(1, []).append((r, -(-r // 64) * ln, r * ln))
        for k in (2, 3, 4, 6):
            cuts = np.linspace(0, ln, k + 1).round().astype(int)
            we = pairs = rr = 0
            for i in range(k):
                a, c = max(cuts[i] - 2, 0), cuts[i + 1]
                if c <= a: continue
                okk = ok & passing(pq, pe[a:c])
                rk = int(okk.sum())
                we += -(-rk // 64) * (c - a)
                pairs += rk * (c - a)
                rr += rk
            res.setdefault(k, []).append((rr, we, pairs))
for k, rows in sorted(res.items()):
    a = np.asarray(rows, np.float64)
    print('k=%d: ray-run pairs per body %.0f  wave-elements per body %.0f  lane-elements per body %.0f  fill %.2f'
          % (k, a[:, 0].sum() / nb, a[:, 1].sum() / nb, a[:, 2].sum() / nb, a[:, 2].sum() / (64 * a[:, 1].sum())))
