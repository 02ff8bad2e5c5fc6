"""CPU check of the cluster tree: hierarchical sum == flat sum (float64)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from tuch_amd import ops
from tuch_amd.synthetic import make_body, random_poses
from oracle import lbs as ol

small = len(sys.argv) > 1
body = make_body(rings=12, segs=10, with_geodesics=False) if small else make_body(with_geodesics=False)
V, F = body.num_verts, body.num_faces
faces = body.faces.astype(np.int64)
t0 = time.time(); t = ops.cluster_tree(faces, V); print('build %.2fs' % (time.time() - t0))
nodes, vidx, sign = t['nodes'], t['vidx'], t['sign']
print('nodes', len(nodes), 'exact_len', t['exact_len'], 'stream', len(vidx), 'leaves', (nodes[:, 3] > 0).sum(),
      'cap mean leaf', nodes[nodes[:, 3] > 0, 1].mean(), 'exact mean', nodes[nodes[:, 3] > 0, 3].mean(), 'F', F)
mt = ol.model_tensors(body)
rp = random_poses(2, seed=3)
verts = ol.smpl_forward(mt, torch.as_tensor(rp[2]), torch.as_tensor(rp[0]), torch.as_tensor(rp[1]))[0].numpy().astype(np.float64)

def tris_of(off, ln):
    idx = []; sg = []
    for p in range(off, off + ln):
        if sign[p] != 0: idx.append((vidx[p - 2], vidx[p - 1], vidx[p])); sg.append(sign[p])
    return np.array(idx, np.int64).reshape(-1, 3), np.array(sg)

def half_angles(q, tri, sg):       # q [Q,3], tri [T,3,3]
    A = tri[None, :, 0] - q[:, None]; B = tri[None, :, 1] - q[:, None]; C = tri[None, :, 2] - q[:, None]
    la, lb, lc = [np.linalg.norm(x, axis=2) for x in (A, B, C)]
    num = np.einsum('qtk,qtk->qt', A, np.cross(B, C))
    den = la * lb * lc + (A * B).sum(2) * lc + (A * C).sum(2) * lb + (B * C).sum(2) * la
    return (np.arctan2(num, den) * sg[None]).sum(1)

# every face exactly once in the exact region
et, es = tris_of(0, t['exact_len'])
canon = lambda a: np.sort(np.stack([np.roll(a, k, 1) for k in range(3)]), 0)
fs = {tuple(sorted(x)) for x in faces}; assert len(et) == F and {tuple(sorted(x)) for x in et} == fs
cache = {}
for b in range(verts.shape[0]):
    vb = verts[b]
    flat = half_angles(vb, vb[faces], np.ones(F)) / (2 * np.pi)
    tree = np.zeros(V); visited_exact = 0; visited_cap = 0
    nb = []
    for i in range(len(nodes)):
        lo, hi = None, None
    # leaf boxes then bottom-up via children (nodes in preorder: children have larger index)
    mn = np.zeros((len(nodes), 3)); mx = np.zeros((len(nodes), 3))
    for i in range(len(nodes) - 1, -1, -1):
        if nodes[i, 3] > 0:
            vs = vidx[nodes[i, 2]:nodes[i, 2] + nodes[i, 3]]; mn[i] = vb[vs].min(0); mx[i] = vb[vs].max(0)
        else:
            c0, c1 = nodes[i, 5], nodes[i, 6]; mn[i] = np.minimum(mn[c0], mn[c1]); mx[i] = np.maximum(mx[c0], mx[c1])
    qp = t['qperm']
    for qb in range(len(qp) // 128):
        q = qp[qb * 128:(qb + 1) * 128]; pts = vb[q]; acc = np.zeros(128)
        node = 0
        while node < len(nodes):
            nd = nodes[node]
            inside = np.all((pts >= mn[node]) & (pts <= mx[node]), axis=1).any()
            if not inside:
                if nd[1] > 0:
                    if ('c', node) not in cache: cache[('c', node)] = tris_of(nd[0], nd[1])
                    ti, sg = cache[('c', node)]; acc += half_angles(pts, vb[ti], sg); visited_cap += nd[1]
                node = nd[4]
            elif nd[3] > 0:
                if ('e', node) not in cache: cache[('e', node)] = tris_of(nd[2], nd[3])
                ti, sg = cache[('e', node)]; acc += half_angles(pts, vb[ti], sg); visited_exact += nd[3]
                node = nd[4]
            else:
                node += 1
        tree[q] = acc / (2 * np.pi)
    print('body', b, 'max |tree - flat| %.3e' % np.abs(tree - flat).max(), 'work ratio %.3f (exact %.3f cap %.3f)' % (
        (visited_exact + visited_cap) / (len(qp) // 128 * t['exact_len']), visited_exact / (len(qp) // 128 * t['exact_len']),
        visited_cap / (len(qp) // 128 * t['exact_len'])))
