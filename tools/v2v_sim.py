import sys, numpy as np, torch
sys.path.insert(0, '.')
from synthetic import make_body, random_poses
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
    vb = verts[b]
    mn, arg = oc.v2v_min_masked(vb, gm)
    P = vb[perm]
    lo = np.stack([P[rows[i, 0]:rows[i, 0] + rows[i, 1]].min(0) for i in leaves]); hi = np.stack([P[rows[i, 0]:rows[i, 0] + rows[i, 1]].max(0) for i in leaves])
    tot_rows = 0; need_rows = 0; masked_rows = 0; nearmask = 0
    dfin = mn[perm]
    print('final min dist: median %.3f p90 %.3f max %.3f (sqrt)' % tuple(np.sqrt(np.percentile(dfin[np.isfinite(dfin)], [50, 90, 100]))))
    gmp = gm[perm][:, perm]
    for qb in range(len(qperm) // 128):
        cols = np.arange(qb * 128, min(V, qb * 128 + 128))
        pc = P[cols]
        e = np.maximum(np.maximum(lo[None] - pc[:, None], pc[:, None] - hi[None]), 0)   # [cols, leaves, 3]
        g = (e ** 2).sum(2)
        can = (g <= dfin[cols][:, None]).any(0)
        for li, i in enumerate(leaves):
            r0, rn = rows[i]
            allmasked = not gmp[r0:r0 + rn][:, cols].any()
            tot_rows += rn
            if allmasked: masked_rows += rn
            elif can[li]: need_rows += rn
    print('body %d: rows needed (ideal pruning) %.3f, statically masked %.3f' % (b, need_rows / tot_rows, masked_rows / tot_rows))
