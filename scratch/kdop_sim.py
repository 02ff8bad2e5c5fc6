import sys, numpy as np, torch
sys.path.insert(0, '.')
from tuch_amd.synthetic import make_body, random_poses
from tuch_amd import ops
from oracle import lbs as ol
body = make_body(with_geodesics=False)
V = body.num_verts; faces = body.faces.astype(np.int64)
t = ops.cluster_tree(faces, V)
nodes, vidx, qperm = t['nodes'], t['vidx'], t['qperm']
mt = ol.model_tensors(body)
rp = random_poses(4, seed=3)
verts = ol.smpl_forward(mt, torch.as_tensor(rp[2]), torch.as_tensor(rp[0]), torch.as_tensor(rp[1]))[0].numpy().astype(np.float64)
A3 = np.eye(3)
A9 = np.concatenate([np.eye(3), np.array([[1, 1, 0], [1, -1, 0], [1, 0, 1], [1, 0, -1], [0, 1, 1], [0, 1, -1]]) / np.sqrt(2)])
A13 = np.concatenate([A9, np.array([[1, 1, 1], [1, 1, -1], [1, -1, 1], [1, -1, -1]]) / np.sqrt(3)])
n = len(nodes)
def nodeverts(i, cache={}):
    if i not in cache:
        if nodes[i, 3] > 0: cache[i] = np.unique(vidx[nodes[i, 2]:nodes[i, 2] + nodes[i, 3]])
        else: cache[i] = np.union1d(nodeverts(nodes[i, 5]), nodeverts(nodes[i, 6]))
    return cache[i]
for name, A in (('aabb', A3), ('9-dop', A9), ('13-dop', A13)):
    tot = 0; totq = 0
    for b in range(2):
        vb = verts[b]; pr = vb @ A.T      # [V, k]
        lo = np.stack([pr[nodeverts(i)].min(0) for i in range(n)]); hi = np.stack([pr[nodeverts(i)].max(0) for i in range(n)])
        for qsize in (128,):
            for qb in range(len(qperm) // 128):
                q = qperm[qb * 128:(qb + 1) * 128]; pq = pr[q]
                node = 0; steps = 0
                while node < n:
                    near = np.all((pq >= lo[node]) & (pq <= hi[node]), axis=1).any()
                    if near and nodes[node, 3] == 0: node += 1; continue
                    steps += nodes[node, 3] if near else nodes[node, 1]
                    node = nodes[node, 4]
                tot += steps; totq += 1
    print(name, 'steps per block %.0f  fraction of exact stream %.3f' % (tot / totq, tot / totq / t['exact_len']))
