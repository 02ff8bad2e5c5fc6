"""Work estimate v2: smoothed clusters, per-query containment test, two levels."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from tuch_amd.synthetic import make_body, random_poses
from oracle import lbs as ol

body = make_body()
V, F = body.num_verts, body.num_faces
faces = body.faces.astype(np.int64)
mt = ol.model_tensors(body)
rp = random_poses(8, seed=3)
verts = ol.smpl_forward(mt, torch.as_tensor(rp[2]), torch.as_tensor(rp[0]), torch.as_tensor(rp[1]))[0].numpy()
tmpl = body.v_template

def kd(points, ids, size, out):
    if len(ids) <= size:
        out.append(ids); return
    p = points[ids]; ax = np.argmax(p.max(0) - p.min(0))
    order = ids[np.argsort(p[:, ax], kind='stable')]
    nleaf = -(-len(ids) // size); half = (nleaf // 2) * size
    kd(points, order[:half], size, out); kd(points, order[half:], size, out)

# face adjacency
ekey = {}
for f in range(F):
    for k in range(3):
        a, b = faces[f, k], faces[f, (k + 1) % 3]
        ekey.setdefault((min(a, b), max(a, b)), []).append(f)
adj = -np.ones((F, 3), np.int64)
for f in range(F):
    for k in range(3):
        a, b = faces[f, k], faces[f, (k + 1) % 3]
        l = ekey[(min(a, b), max(a, b))]
        adj[f, k] = l[0] if l[1] == f else l[1]

def smooth(label, iters=20):
    label = label.copy()
    for _ in range(iters):
        moved = 0
        for f in range(F):
            nl = label[adj[f]]
            other = nl[nl != label[f]]
            if len(other) >= 2:
                vals, cnt = np.unique(other, return_counts=True)
                if cnt.max() >= 2:
                    label[f] = vals[np.argmax(cnt)]; moved += 1
        if moved == 0: break
    return label

def nboundary(label, c):
    fs = np.where(label == c)[0]
    return int((label[adj[fs]] != c).sum())

def run(K, QB, K2=None):
    cen = tmpl[faces].mean(1); cl = []; kd(cen, np.arange(F), K, cl)
    label = np.empty(F, np.int64)
    for i, c in enumerate(cl): label[c] = i
    nb0 = np.mean([nboundary(label, i) for i in range(len(cl))])
    label = smooth(label)
    ncl = len(cl)
    nb = np.array([nboundary(label, i) for i in range(ncl)]); nk = np.array([(label == i).sum() for i in range(ncl)])
    qblocks = []; kd(tmpl, np.arange(V), QB, qblocks)
    res = {'blk_aabb': 0.0, 'anyq_aabb': 0.0, 'anyq_aabb_sph': 0.0, 'perq': 0.0}
    for b in range(verts.shape[0]):
        vb = verts[b]
        cv = [np.unique(faces[label == i]) for i in range(ncl)]
        cmin = np.stack([vb[c].min(0) for c in cv]); cmax = np.stack([vb[c].max(0) for c in cv])
        cc = 0.5 * (cmin + cmax); cr = np.array([np.linalg.norm(vb[c] - cc[i], axis=1).max() for i, c in enumerate(cv)])
        inside = np.all((vb[:, None] >= cmin[None]) & (vb[:, None] <= cmax[None]), axis=2)       # [V, nc]
        insph = inside & (np.linalg.norm(vb[:, None] - cc[None], axis=2) <= cr[None])
        wn, wf = nk * 1.15 + 2, nb + 2
        res['perq'] += np.where(insph, wn[None], wf[None]).sum() / (V * F * 1.15)
        for name, m in (('anyq_aabb', inside), ('anyq_aabb_sph', insph)):
            near = np.stack([m[q].any(0) for q in qblocks])
            res[name] += np.where(near, wn[None], wf[None]).sum() / (len(qblocks) * F * 1.15)
        qmin = np.stack([vb[q].min(0) for q in qblocks]); qmax = np.stack([vb[q].max(0) for q in qblocks])
        near = np.all((qmin[:, None] <= cmax[None]) & (qmax[:, None] >= cmin[None]), axis=2)
        res['blk_aabb'] += np.where(near, wn[None], wf[None]).sum() / (len(qblocks) * F * 1.15)
    print('K %d QB %d: clusters %d boundary %.1f -> %.1f; work:' % (K, QB, ncl, nb0, nb.mean()),
          {k: round(v / verts.shape[0], 3) for k, v in res.items()})

for K in (64, 128, 256):
    for QB in (64, 128):
        run(K, QB)
