"""Host-side checks of the face-cluster tree behind the hierarchical winding numbers
(tuch_amd/csrc/cluster_tree.hip).  No GPU: the tree is built by the C ABI on the host and walked here
in float64 with the reference's solid-angle formula (tuch/utils/contact.py:49-109); the walk must
reproduce the plain sum over all faces (contact.py:112-147) to rounding."""
from __future__ import annotations

import numpy as np
import pytest

from helpers import golden
from oracle import contact as oc
from tuch_amd import ops


def triangles_of(vidx, sign, off, length):
    idx, sg = [], []
    for p in range(off, off + length):
        if sign[p] != 0:
            idx.append((vidx[p - 2], vidx[p - 1], vidx[p]))
            sg.append(sign[p])
    return np.array(idx, np.int64).reshape(-1, 3), np.array(sg, np.float64)


def half_angle_sum(q, tri, sg):
    """sum_f sign_f * atan2(A.(BxC), |A||B||C| + (A.B)|C| + (A.C)|B| + (B.C)|A|) for q [Q,3], tri [T,3,3]."""
    if len(sg) == 0:
        return np.zeros(len(q))
    a, b, c = (tri[None, :, k] - q[:, None] for k in range(3))
    la, lb, lc = (np.linalg.norm(x, axis=2) for x in (a, b, c))
    num = np.einsum('qtk,qtk->qt', a, np.cross(b, c))
    den = la * lb * lc + (a * b).sum(2) * lc + (a * c).sum(2) * lb + (b * c).sum(2) * la
    return (np.arctan2(num, den) * sg[None]).sum(1)


def walk(tree, verts, start=0, stop=None, stats=None):
    """Winding numbers of all vertices by the tree walk the device performs (wave = 128 queries)."""
    nodes, vidx, sign, qperm = tree['nodes'], tree['vidx'], tree['sign'], tree['qperm']
    n = len(nodes)
    lo, hi = np.zeros((n, 3)), np.zeros((n, 3))
    for i in range(n - 1, -1, -1):            # preorder: children have larger indices
        if nodes[i, 3] > 0:
            vs = vidx[nodes[i, 2]:nodes[i, 2] + nodes[i, 3]]
            lo[i], hi[i] = verts[vs].min(0), verts[vs].max(0)
        else:
            lo[i] = np.minimum(lo[nodes[i, 5]], lo[nodes[i, 6]])
            hi[i] = np.maximum(hi[nodes[i, 5]], hi[nodes[i, 6]])
    out = np.zeros(len(verts))
    cache = {}
    for qb in range(len(qperm) // 128):
        q = qperm[qb * 128:(qb + 1) * 128]
        pts = verts[q]
        acc = np.zeros(128)
        node, end = start, (nodes[start, 4] if stop is None else stop)
        while node < end:
            nd = nodes[node]
            near = np.all((pts >= lo[node]) & (pts <= hi[node]), axis=1).any()
            if near and nd[3] == 0:
                node += 1
                continue
            off, length = (nd[2], nd[3]) if near else (nd[0], nd[1])
            if (off, length) not in cache:
                cache[(off, length)] = triangles_of(vidx, sign, off, length)
            ti, sg = cache[(off, length)]
            acc += half_angle_sum(pts, verts[ti], sg)
            if stats is not None:
                stats['near' if near else 'far'] += length
            node = nd[4]
        out[q] = acc / (2 * np.pi)
    return out


@pytest.mark.parametrize('tag', ['small', 'medium', 'ico_small', 'ico_medium'])
def test_tree_structure(tag):
    g = golden(tag)
    faces, v = g['faces'], g['verts'].shape[1]
    t = ops.cluster_tree(faces, v)
    nodes, vidx, sign = t['nodes'], t['vidx'], t['sign']
    n = len(nodes)
    assert nodes[0, 4] == n and nodes[0, 1] == 0          # the root spans everything and has no boundary
    assert nodes[0, 7] == len(faces)
    # every face exactly once, with its orientation, in the leaf region of the stream
    tri, sg = triangles_of(vidx, sign, 0, t['exact_len'])
    assert len(tri) == len(faces)
    want = {tuple(int(x) for x in np.roll(f, -int(np.argmin(f)))) for f in faces}
    got = set()
    for (a, b, c), s in zip(tri, sg):
        f = (a, b, c) if s > 0 else (a, c, b)
        got.add(tuple(int(x) for x in np.roll(f, -int(np.argmin(f)))))
    assert got == want
    leaf_faces = 0
    for i in range(n):
        cap_off, cap_len, ex_off, ex_len, skip, c0, c1, nf = nodes[i]
        assert cap_len % 3 == 0 and ex_len % 3 == 0 and i < skip <= n
        if c0 >= 0:
            assert ex_len == 0 and c0 == i + 1 and nodes[c0, 4] == c1 and nodes[c1, 4] == skip
            assert nf == nodes[c0, 7] + nodes[c1, 7]
        else:
            assert c1 < 0 and skip == i + 1 and ex_len > 0
            assert len(triangles_of(vidx, sign, ex_off, ex_len)[1]) == nf
            leaf_faces += nf
        if cap_len:
            assert cap_off >= t['exact_len']
    assert leaf_faces == len(faces)
    # the query order is a permutation of the vertices (padded with repeats)
    assert set(t['qperm'].tolist()) == set(range(v))
    # every frontier covers the mesh; every launch order lists each (subtree, block) pair once
    qblocks = len(t['qperm']) // 128
    for f in range(len(t['frontier_off']) - 1):
        fr = t['frontier_nodes'][t['frontier_off'][f]:t['frontier_off'][f + 1]]
        assert sum(nodes[i, 7] for i in fr) == len(faces)
        order = t['launch_order'][t['frontier_off'][f] * qblocks:t['frontier_off'][f + 1] * qblocks]
        assert sorted(order.tolist()) == sorted((s << 16) | q for s in range(len(fr)) for q in range(qblocks))


@pytest.mark.parametrize('tag', ['small', 'medium', 'ico_small', 'ico_medium'])
def test_cap_of_every_node_equals_its_faces_for_outside_queries(tag):
    """The identity the method rests on, node by node: for queries outside the node's box, the cap
    triangulation subtends the same solid angle as the node's faces."""
    g = golden(tag)
    faces = g['faces']
    verts = g['verts'][0].astype(np.float64)
    t = ops.cluster_tree(faces, verts.shape[0], leaf_faces=16)
    nodes, vidx, sign = t['nodes'], t['vidx'], t['sign']
    checked = 0
    for i in range(1, len(nodes)):
        leaves = [j for j in range(i, nodes[i, 4]) if nodes[j, 3] > 0]
        parts = [triangles_of(vidx, sign, nodes[j, 2], nodes[j, 3]) for j in leaves]
        ti, sg = np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        vs = np.unique(ti)
        outside = ~np.all((verts >= verts[vs].min(0)) & (verts <= verts[vs].max(0)), axis=1)
        if not outside.any():
            continue
        ci, cs = triangles_of(vidx, sign, nodes[i, 0], nodes[i, 1])
        q = verts[outside]
        np.testing.assert_allclose(half_angle_sum(q, verts[ci], cs), half_angle_sum(q, verts[ti], sg), rtol=0, atol=1e-11)
        checked += 1
    assert checked >= (len(nodes) - 1) // 2


@pytest.mark.parametrize('tag,leaf', [('small', 16), ('small', 64), ('medium', 24), ('medium', 64), ('ico_small', 16),
                                      ('ico_medium', 24), ('ico_medium', 32), ('ico_medium', 64)])
def test_tree_walk_equals_flat_sum(tag, leaf):
    g = golden(tag)
    faces = g['faces']
    t = ops.cluster_tree(faces, g['verts'].shape[1], leaf_faces=leaf)
    for b in range(g['verts'].shape[0]):
        verts = g['verts'][b].astype(np.float64)
        stats = {'near': 0, 'far': 0}
        w = walk(t, verts, stats=stats)
        flat = half_angle_sum(verts, verts[faces], np.ones(len(faces))) / (2 * np.pi)
        np.testing.assert_allclose(w, flat, rtol=0, atol=1e-11)
        # and against the committed reference output (float32 arithmetic there)
        assert np.abs(w - g['winding'][b]).max() < 2e-4
        if tag == 'medium':
            assert stats['far'] > 0           # the caps were actually exercised (small: one block holds every vertex)
    # the walk restricted to the subtrees of a frontier adds up to the same numbers
    fo = t['frontier_off']
    f = len(fo) - 2
    verts = g['verts'][0].astype(np.float64)
    total = sum(walk(t, verts, start=int(s)) for s in t['frontier_nodes'][fo[f]:fo[f + 1]])
    np.testing.assert_allclose(total, walk(t, verts), rtol=0, atol=1e-11)


def test_open_or_inconsistent_meshes_are_rejected():
    g = golden('small')
    faces = g['faces'].copy()
    with pytest.raises(RuntimeError):
        ops.cluster_tree(faces[:-1], g['verts'].shape[1])            # a hole
    flipped = faces.copy()
    flipped[0] = flipped[0][::-1]
    with pytest.raises(RuntimeError):
        ops.cluster_tree(flipped, g['verts'].shape[1])               # inconsistent orientation


def test_tree_is_deterministic():
    g = golden('medium')
    a = ops.cluster_tree(g['faces'], g['verts'].shape[1])
    b = ops.cluster_tree(g['faces'], g['verts'].shape[1])
    for k in a:
        assert np.array_equal(a[k], b[k])
