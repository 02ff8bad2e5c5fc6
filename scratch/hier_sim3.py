"""Work estimate v3: topology-only clustering (FPS Voronoi by hop distance), pairwise merge tree, full traversal."""
import sys, numpy as np, torch, collections
sys.path.insert(0, '.')
from tuch_amd.synthetic import make_body, random_poses
from oracle import lbs as ol

body = make_body()
V, F = body.num_verts, body.num_faces
faces = body.faces.astype(np.int64)
mt = ol.model_tensors(body)
NB = 8
rp = random_poses(NB, seed=3)
verts = ol.smpl_forward(mt, torch.as_tensor(rp[2]), torch.as_tensor(rp[0]), torch.as_tensor(rp[1]))[0].numpy()

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

def bfs(sources):
    dist = np.full(F, -1, np.int64); owner = np.full(F, -1, np.int64)
    dq = collections.deque()
    for i, s in enumerate(sources):
        dist[s] = 0; owner[s] = i; dq.append(s)
    while dq:
        f = dq.popleft()
        for g in adj[f]:
            if dist[g] < 0:
                dist[g] = dist[f] + 1; owner[g] = owner[f]; dq.append(g)
    return dist, owner

def voronoi(k):
    seeds = [0]
    dist, _ = bfs(seeds)
    seeds = [int(np.argmax(dist))]                 # pseudo-peripheral start
    dmin, _ = bfs(seeds)
    while len(seeds) < k:
        s = int(np.argmax(dmin)); seeds.append(s)
        d, _ = bfs([s]); dmin = np.minimum(dmin, d)
    _, owner = bfs(seeds)
    return owner

def smooth(label, iters=20):
    label = label.copy()
    for _ in range(iters):
        moved = 0
        for f in range(F):
            nl = label[adj[f]]; other = nl[nl != label[f]]
            if len(other) >= 2:
                vals, cnt = np.unique(other, return_counts=True)
                if cnt.max() >= 2:
                    label[f] = vals[np.argmax(cnt)]; moved += 1
        if moved == 0: break
    return label

def build_tree(label):
    """returns list of nodes: dict(faces=set array, children=[...])"""
    ncl = label.max() + 1
    nodes = [{'faces': np.where(label == i)[0], 'children': []} for i in range(ncl)]
    cur = list(range(ncl)); lab = label.copy()
    while len(cur) > 1:
        # adjacency weights between current groups
        w = collections.defaultdict(int)
        for f in range(F):
            for g in adj[f]:
                if lab[f] != lab[g]: w[(lab[f], lab[g])] += 1
        size = {c: len(nodes[c]['faces']) for c in cur}
        matched = {}; order = sorted(cur, key=lambda c: size[c])
        for c in order:
            if c in matched: continue
            best, bw = None, -1
            for d in cur:
                if d != c and d not in matched and w.get((c, d), 0) > 0:
                    score = w[(c, d)] / np.sqrt(size[d])
                    if score > bw: best, bw = d, score
            if best is not None:
                matched[c] = best; matched[best] = c
        new = []; done = set()
        for c in cur:
            if c in done: continue
            if c in matched:
                d = matched[c]; done.add(c); done.add(d)
                nodes.append({'faces': np.concatenate([nodes[c]['faces'], nodes[d]['faces']]), 'children': [c, d]})
                n = len(nodes) - 1; lab[nodes[n]['faces']] = n; new.append(n)
            else:
                new.append(c)
        if len(new) == len(cur): break
        cur = new
    return nodes, cur

def nboundary(fs):
    inset = np.zeros(F, bool); inset[fs] = True
    return int((~inset[adj[fs]]).sum())

for K in (48, 64, 96, 128):
    label = smooth(voronoi(int(round(F / K))))
    nodes, roots = build_tree(label)
    for n in nodes:
        n['nb'] = nboundary(n['faces']); n['verts'] = np.unique(faces[n['faces']])
    leaves = [n for n in nodes if not n['children']]
    print('K %d: leaves %d size mean %.1f max %d, boundary mean %.1f; roots %d' % (K, len(leaves), np.mean([len(n['faces']) for n in leaves]),
          max(len(n['faces']) for n in leaves), np.mean([n['nb'] for n in leaves]), len(roots)))
    # preorder of leaves for vertex ordering
    order = []
    def pre(i):
        if not nodes[i]['children']: order.append(i)
        for c in nodes[i]['children']: pre(c)
    for r in roots: pre(r)
    leafpos = {l: i for i, l in enumerate(order)}
    vleaf = np.full(V, 10**9)
    for l in order:
        vs = nodes[l]['verts']; vleaf[vs] = np.minimum(vleaf[vs], leafpos[l])
    vorder = np.argsort(vleaf, kind='stable')
    for QB in (64, 128):
        qblocks = [vorder[i:i + QB] for i in range(0, V, QB)]
        tot = 0.0; capw = 0.0
        for b in range(NB):
            vb = verts[b]
            for n in nodes:
                n['min'] = vb[n['verts']].min(0); n['max'] = vb[n['verts']].max(0)
            for q in qblocks:
                pts = vb[q]
                def trav(i):
                    n = nodes[i]
                    inside = np.all((pts >= n['min']) & (pts <= n['max']), axis=1).any()
                    if not inside: return n['nb'] + 2, n['nb'] + 2
                    if not n['children']: return len(n['faces']) * 1.15 + 2, 0
                    a = [trav(c) for c in n['children']]
                    return sum(x[0] for x in a), sum(x[1] for x in a)
                for r in roots:
                    w, c = trav(r); tot += w; capw += c
        denom = NB * len(qblocks) * F * 1.15
        print('   QB %d: work ratio %.3f (cap share %.3f)' % (QB, tot / denom, capw / denom))
