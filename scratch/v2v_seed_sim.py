import sys, numpy as np, torch
sys.path.insert(0, '.')
from tuch_amd.synthetic import make_body, random_poses
from tuch_amd import ops
from oracle import lbs as ol, contact as oc
body = make_body()
V = body.num_verts; faces = body.faces.astype(np.int64)
t = ops.cluster_tree(faces, V)
nodes, rows, qperm = t['nodes'], t['rows'], t['qperm']
gm = body.geodesics > 0.3
mt = ol.model_tensors(body)
rp = random_poses(4, seed=3)
verts = ol.smpl_forward(mt, torch.as_tensor(rp[2]), torch.as_tensor(rp[0]), torch.as_tensor(rp[1]))[0].numpy()
leaves = [i for i in range(len(nodes)) if nodes[i, 5] < 0 and rows[i, 1] > 0]
perm = qperm[:V]
for b in range(2):
    vb = verts[b]; P = vb[perm]; gmp = gm[perm][:, perm]
    D = None
    lo = np.stack([P[rows[i, 0]:rows[i, 0] + rows[i, 1]].min(0) for i in leaves]); hi = np.stack([P[rows[i, 0]:rows[i, 0] + rows[i, 1]].max(0) for i in leaves])
    rn = np.array([rows[i, 1] for i in leaves])
    tot = need_ideal = need_seed1 = need_seed3 = 0
    for qb in range(0, V, 64):
        cols = np.arange(qb, min(V, qb + 64)); pc = P[cols]
        d = ((pc[:, None] - P[None]) ** 2).sum(2); d = np.where(gmp[:, cols].T, d, np.inf)      # [cols, rows] allowed = gm[row][col]
        fin = d.min(1)
        e = np.maximum(np.maximum(lo[None] - pc[:, None], pc[:, None] - hi[None]), 0); g = (e ** 2).sum(2)   # [cols, leaves]
        blo, bhi = pc.min(0), pc.max(0)
        eg = np.maximum(np.maximum(lo - bhi, blo - hi), 0); gap = (eg ** 2).sum(1)
        adm = np.array([np.isfinite(d[:, rows[i, 0]:rows[i, 0] + rows[i, 1]]).any() for i in leaves])
        order = np.argsort(np.where(adm, gap, np.inf))
        def bound(k):
            bb = np.full(len(cols), np.inf)
            for li in order[:k]:
                i = leaves[li]; bb = np.minimum(bb, d[:, rows[i, 0]:rows[i, 0] + rows[i, 1]].min(1))
            return bb
        for name, bd in (('ideal', fin), ('seed1', bound(1)), ('seed3', bound(3))):
            keep = adm & (g <= bd[:, None]).any(0)
            if name == 'ideal': need_ideal += rn[keep].sum()
            elif name == 'seed1': need_seed1 += rn[keep].sum()
            else: need_seed3 += rn[keep].sum()
        tot += rn.sum()
    print('body %d: rows evaluated / all: ideal %.3f, static with 1-leaf seed %.3f, with 3-leaf seed %.3f' % (b, need_ideal / tot, need_seed1 / tot, need_seed3 / tot))
