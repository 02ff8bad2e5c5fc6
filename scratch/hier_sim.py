"""Work estimate for boundary-cap (exact hierarchical) winding numbers."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from tuch_amd.synthetic import make_body, random_poses
from oracle import lbs as ol

body = make_body()
V, F = body.num_verts, body.num_faces
faces = body.faces.astype(np.int64)
mt = ol.model_tensors(body)
rp = random_poses(8, seed=3)
print(type(rp), [getattr(x, 'shape', None) for x in rp] if isinstance(rp, (tuple, list)) else rp.keys())
if isinstance(rp, dict):
    betas, body_pose, go = rp['betas'], rp['body_pose'], rp['global_orient']
else:
    body_pose, go, betas = rp[0], rp[1], rp[2]
out = ol.smpl_forward(mt, torch.as_tensor(betas), torch.as_tensor(body_pose), torch.as_tensor(go))
verts = (out[0] if isinstance(out, (tuple, list)) else out['vertices']).numpy()
print('verts', verts.shape)
tmpl = body.v_template

def kd(points, ids, size, out):
    if len(ids) <= size:
        out.append(ids); return
    p = points[ids]; ax = np.argmax(p.max(0) - p.min(0))
    order = ids[np.argsort(p[:, ax], kind='stable')]
    # split at a multiple of size so that leaves are full
    nleaf = -(-len(ids) // size); half = (nleaf // 2) * size
    kd(points, order[:half], size, out); kd(points, order[half:], size, out)

def clusters_for(K):
    cen = tmpl[faces].mean(1); out = []
    kd(cen, np.arange(F), K, out); return out

def boundary_edges(cl):
    f = faces[cl]
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key = np.minimum(e[:, 0], e[:, 1]) * V + np.maximum(e[:, 0], e[:, 1])
    u, c = np.unique(key, return_counts=True)
    return int((c == 1).sum())

for QB in (128, 256):
    qblocks = []; kd(tmpl, np.arange(V), QB, qblocks)
    for K in (64, 128, 256, 512):
        cls = clusters_for(K)
        nb = np.array([boundary_edges(c) for c in cls]); nk = np.array([len(c) for c in cls])
        tot = 0.0; near_frac = 0.0
        for b in range(verts.shape[0]):
            vb = verts[b]
            cmin = np.stack([vb[faces[c]].reshape(-1, 3).min(0) for c in cls]); cmax = np.stack([vb[faces[c]].reshape(-1, 3).max(0) for c in cls])
            qmin = np.stack([vb[q].min(0) for q in qblocks]); qmax = np.stack([vb[q].max(0) for q in qblocks])
            near = np.all((qmin[:, None] <= cmax[None]) & (qmax[:, None] >= cmin[None]), axis=2)   # [nq, nc]
            work = np.where(near, nk[None] * 1.15 + 2, nb[None] + 2)
            tot += work.sum() / (len(qblocks) * F * 1.15); near_frac += near.mean()
        print('QB %d K %d: clusters %d mean boundary %.1f  near %.3f  work ratio %.3f' % (QB, K, len(cls), nb.mean(), near_frac / verts.shape[0], tot / verts.shape[0]))
