"""Synthetic SMPL-shaped assets.

The licensed SMPL files the reference loads (configs/config.py:74-87: model .pkl,
geodesic matrix, segment .ply + segm_utils, HD regressor, GMM prior, extra joint
regressor; DSC classes.pkl / ContactSigSMPL.pkl) are absent, so every test and
bench input is generated here, deterministically, with NumPy PCG64 only.

What is produced has the *shapes and roles* of the real assets:

* a closed, consistently oriented genus-0 mesh with the UV-sphere topology
  ``rings x segs + 2`` vertices -> exactly V=6890, F=13776 at (84, 82);
* a 24-joint SMPL kinematic tree, skinning weights, shape/pose blend shapes,
  joint regressors, 21 picked vertices + 9 extra regressed joints and a
  49-entry joint map (tuch/models/smpl.py:37-49);
* graph-geodesic distances on the template (role of
  smpl_neutral_geodesic_dist.npy, demo_smplify_dc.py:63);
* contact regions + region pairs (role of ContactSigSMPL.pkl / classes.pkl,
  tuch/train/train_module.py:64-66);
* body segments with ordered boundary loops (role of the segment .ply files and
  segm_utils.segments, tuch/utils/segmentation.py:40-47);
* an HD barycentric vertex regressor (role of smpl_neutral_hd_vert_regressor.npy
  and faces_vert_is_sampled_from, tuch/train/loss.py:81-88);
* an 8-component, 69-D GMM pose prior (role of gmm_08.pkl, tuch/smplify/prior.py).

None of this is anatomical.  The body is a star-shaped "starfish person": five
capsules (head/torso-up, two arms, two legs) radiating from the pelvis, which
is enough for limbs to swing into each other and self-penetrate.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Tuple

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import dijkstra

SMPL_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
    dtype=np.int64)

# limb id, direction, length, radius (metres).  0 = up (spine/neck/head),
# 1/2 = left/right arm, 3/4 = left/right leg.
_LIMBS = [
    ((0.0, 1.0, 0.0), 0.52, 0.12),
    ((0.93, 0.36, 0.0), 0.58, 0.07),
    ((-0.93, 0.36, 0.0), 0.58, 0.07),
    ((0.25, -0.968, 0.0), 0.72, 0.095),
    ((-0.25, -0.968, 0.0), 0.72, 0.095),
]
_TORSO_RADIUS = 0.19

# joint -> (limb, fraction along the limb axis); pelvis sits at the origin.
_JOINT_PLACEMENT = {
    0: (0, 0.0), 3: (0, 0.14), 6: (0, 0.28), 9: (0, 0.42), 12: (0, 0.62), 15: (0, 0.80),
    13: (1, 0.10), 16: (1, 0.22), 18: (1, 0.55), 20: (1, 0.85), 22: (1, 0.95),
    14: (2, 0.10), 17: (2, 0.22), 19: (2, 0.55), 21: (2, 0.85), 23: (2, 0.95),
    1: (3, 0.12), 4: (3, 0.52), 7: (3, 0.88), 10: (3, 0.96),
    2: (4, 0.12), 5: (4, 0.52), 8: (4, 0.88), 11: (4, 0.96),
}


@dataclasses.dataclass
class SyntheticBody:
    """All constants a caller of the hot path needs, as NumPy arrays."""
    v_template: np.ndarray          # [V,3] f32
    faces: np.ndarray               # [F,3] int64, outward oriented
    shapedirs: np.ndarray           # [V,3,10] f32
    posedirs: np.ndarray            # [207, V*3] f32
    J_regressor: np.ndarray         # [24,V] f32
    lbs_weights: np.ndarray         # [V,24] f32
    parents: np.ndarray             # [24] int64
    extra_vertex_ids: np.ndarray    # [21] int64 (role of smplx VertexJointSelector)
    J_regressor_extra: np.ndarray   # [9,V] f32
    joint_map: np.ndarray           # [49] int64 into the 54 joints
    geodesics: Optional[np.ndarray]  # [V,V] f32 (None if skipped)
    regions: Dict[str, np.ndarray]  # role of csig: name -> vertex ids
    region_pairs: List[Tuple[str, str]]  # role of classes
    segments: Dict[str, dict]       # name -> {'vidx': ids, 'bands': {band: ordered loop ids}}
    hd_regressor: np.ndarray        # [N_hd, V] dense f32 is too big at full size: stored sparse
    hd_bary_idx: np.ndarray         # [N_hd,3] int64 vertex ids
    hd_bary_w: np.ndarray           # [N_hd,3] f32 barycentric weights
    hd_face_id: np.ndarray          # [N_hd] int64 (role of faces_vert_is_sampled_from)
    gmm: dict                       # {'means','covars','weights'}
    limb_of_vertex: np.ndarray      # [V] int64
    axial_of_vertex: np.ndarray     # [V] f32

    @property
    def num_verts(self) -> int:
        return int(self.v_template.shape[0])

    @property
    def num_faces(self) -> int:
        return int(self.faces.shape[0])


# --------------------------------------------------------------------------- mesh
def _uv_sphere_topology(rings: int, segs: int) -> Tuple[np.ndarray, np.ndarray]:
    """Unit directions and faces.  Vertex 0 = north pole, then rings, last = south pole."""
    theta = np.pi * (np.arange(rings) + 1.0) / (rings + 1.0)
    phi = 2.0 * np.pi * np.arange(segs) / segs
    st, ct = np.sin(theta)[:, None], np.cos(theta)[:, None]
    ring_dirs = np.stack([st * np.cos(phi)[None], np.broadcast_to(ct, (rings, segs)),
                          st * np.sin(phi)[None]], -1).reshape(-1, 3)
    dirs = np.concatenate([[[0.0, 1.0, 0.0]], ring_dirs, [[0.0, -1.0, 0.0]]], 0)
    vid = lambda r, s: 1 + r * segs + (s % segs)
    faces = []
    south = 1 + rings * segs
    for s in range(segs):
        faces.append((0, vid(0, s + 1), vid(0, s)))
    for r in range(rings - 1):
        for s in range(segs):
            a, b = vid(r, s), vid(r, s + 1)
            c, d = vid(r + 1, s), vid(r + 1, s + 1)
            faces.append((a, b, d))
            faces.append((a, d, c))
    for s in range(segs):
        faces.append((south, vid(rings - 1, s), vid(rings - 1, s + 1)))
    faces = np.asarray(faces, dtype=np.int64)
    return dirs, faces


def _ico_topology(freq: int) -> Tuple[np.ndarray, np.ndarray]:
    """Geodesic icosahedron of frequency ``freq``: V = 10 freq^2 + 2 (never a multiple of 64), F = 20 freq^2,
    valence 6 except at the twelve corners (5).  Unit directions and faces."""
    t = (1.0 + np.sqrt(5.0)) / 2.0
    corners = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                        [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    corners /= np.linalg.norm(corners, axis=1, keepdims=True)
    base = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
            (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
            (8, 6, 7), (9, 8, 1)]
    index: Dict[tuple, int] = {}
    pts: List[np.ndarray] = []

    def vid(p: np.ndarray) -> int:
        key = tuple(np.round(p * 1e6).astype(np.int64).tolist())
        if key not in index:
            index[key] = len(pts)
            pts.append(p)
        return index[key]
    faces = []
    for a, b, c in base:
        A, B, C = corners[a], corners[b], corners[c]
        grid = {}
        for i in range(freq + 1):
            for j in range(freq + 1 - i):
                p = (A * (freq - i - j) + B * i + C * j) / freq
                grid[(i, j)] = vid(p / np.linalg.norm(p))
        for i in range(freq):
            for j in range(freq - i):
                faces.append((grid[(i, j)], grid[(i + 1, j)], grid[(i, j + 1)]))
                if i + j < freq - 1:
                    faces.append((grid[(i + 1, j)], grid[(i + 1, j + 1)], grid[(i, j + 1)]))
    return np.asarray(pts), np.asarray(faces, dtype=np.int64)


def _random_edge_flips(dirs: np.ndarray, faces: np.ndarray, flips: int, rng, min_valence: int = 4,
                       max_valence: int = 9) -> np.ndarray:
    """``flips`` random edge flips on a closed triangulation of the sphere: the result is still a closed, consistently
    oriented manifold (a flip is taken only if the new diagonal does not exist yet and the quad is convex on the
    sphere) with valences in [min_valence, max_valence] -- the spread of the SMPL mesh (3 ... 10,
    tuch/models/smpl.py:37-42 loads it as it is) instead of a regular grid."""
    faces = faces.copy()
    # orient consistently outward first (the flip's convexity test assumes it)
    vol = np.einsum('ij,ij->i', dirs[faces[:, 0]], np.cross(dirs[faces[:, 1]], dirs[faces[:, 2]]))
    faces[vol < 0] = faces[vol < 0][:, [0, 2, 1]]
    edge: Dict[tuple, list] = {}
    for f, (a, b, c) in enumerate(faces.tolist()):
        for x, y in ((a, b), (b, c), (c, a)):
            edge.setdefault((min(x, y), max(x, y)), []).append(f)
    valence = np.bincount(faces.ravel(), minlength=len(dirs))
    det = lambda x, y, z: float(dirs[x] @ np.cross(dirs[y], dirs[z]))
    done = tries = 0
    keys = sorted(edge)
    while done < flips and tries < 200 * max(flips, 1):
        tries += 1
        a, b = keys[int(rng.integers(len(keys)))]
        if (a, b) not in edge:
            continue
        f1, f2 = edge[(a, b)]
        t1 = faces[f1].tolist()
        k = t1.index(a)
        if t1[(k + 1) % 3] != b:            # make f1 the face that holds the directed edge a -> b
            f1, f2 = f2, f1
            t1 = faces[f1].tolist()
            k = t1.index(a)
        c = t1[(k + 2) % 3]
        t2 = faces[f2].tolist()
        d = t2[(t2.index(b) + 2) % 3]
        if (min(c, d), max(c, d)) in edge or valence[a] <= min_valence or valence[b] <= min_valence \
                or valence[c] >= max_valence or valence[d] >= max_valence:
            continue
        if det(a, d, c) <= 1e-9 or det(d, b, c) <= 1e-9:
            continue
        faces[f1] = (a, d, c)
        faces[f2] = (d, b, c)
        del edge[(a, b)]
        edge[(min(c, d), max(c, d))] = [f1, f2]
        e = edge[(min(a, d), max(a, d))]
        e[e.index(f2)] = f1
        e = edge[(min(b, c), max(b, c))]
        e[e.index(f1)] = f2
        valence[[a, b]] -= 1
        valence[[c, d]] += 1
        done += 1
    return faces


def _limb_axes() -> List[Tuple[np.ndarray, float, float]]:
    out = []
    for d, length, rad in _LIMBS:
        d = np.asarray(d, dtype=np.float64)
        out.append((d / np.linalg.norm(d), float(length), float(rad)))
    return out


def _radius(dirs: np.ndarray) -> np.ndarray:
    """Distance from the origin to the starfish surface along unit directions."""
    p = 10.0
    acc = np.full(dirs.shape[0], _TORSO_RADIUS ** p)
    for d, length, rad in _limb_axes():
        c = dirs @ d
        s2 = np.maximum(1.0 - c * c, 1e-12)
        t_side = rad / np.sqrt(s2)
        disc = np.maximum(length * length * c * c - length * length + rad * rad, 0.0)
        t_cap = length * c + np.sqrt(disc)
        t = np.where(c <= 0.0, rad, np.where(t_side * c <= length, t_side, t_cap))
        acc += t ** p
    return acc ** (1.0 / p)


def _orient_outward(verts: np.ndarray, faces: np.ndarray) -> np.ndarray:
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    vol = np.einsum('ij,ij->i', a, np.cross(b, c))
    if vol.sum() < 0:
        faces = faces[:, [0, 2, 1]]
        vol = -vol
    if not np.all(vol > 0):
        raise RuntimeError('synthetic mesh is not star-shaped/valid (%d bad faces)' % int((vol <= 0).sum()))
    return faces


def _adjacency(num_verts: int, faces: np.ndarray) -> sp.csr_matrix:
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0)
    e = np.concatenate([e, e[:, ::-1]], 0)
    a = sp.coo_matrix((np.ones(len(e)), (e[:, 0], e[:, 1])), shape=(num_verts, num_verts)).tocsr()
    a.data[:] = 1.0
    return a


def _relax(dirs: np.ndarray, faces: np.ndarray, iters: int) -> np.ndarray:
    """Area-weighted smoothing on the surface: every vertex moves toward the
    area-weighted mean of its incident triangles' centroids, then is projected
    back radially.  Equalises triangle areas over the (radially stretched) limbs."""
    num_verts, num_faces = len(dirs), len(faces)
    inc = sp.coo_matrix((np.ones(3 * num_faces),
                         (np.concatenate([faces[:, 0], faces[:, 1], faces[:, 2]]),
                          np.tile(np.arange(num_faces), 3))), shape=(num_verts, num_faces)).tocsr()
    for _ in range(iters):
        pts = dirs * _radius(dirs)[:, None]
        a, b, c = pts[faces[:, 0]], pts[faces[:, 1]], pts[faces[:, 2]]
        area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
        target = (inc @ ((a + b + c) / 3.0 * area[:, None])) / (inc @ area)[:, None]
        new = 0.5 * pts + 0.5 * target
        dirs = new / np.linalg.norm(new, axis=1, keepdims=True)
    return dirs


def _smooth(field: np.ndarray, adj: sp.csr_matrix, iters: int) -> np.ndarray:
    deg = np.asarray(adj.sum(1)).ravel()[:, None]
    for _ in range(iters):
        field = 0.5 * field + 0.5 * (adj @ field) / deg
    return field


# ----------------------------------------------------------------------- skeleton
def _joint_positions() -> np.ndarray:
    axes = _limb_axes()
    j = np.zeros((24, 3))
    for k, (limb, frac) in _JOINT_PLACEMENT.items():
        d, length, _ = axes[limb]
        j[k] = d * (frac * length)
    return j


def _point_segment_distance(p: np.ndarray, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    ab = b - a
    denom = float(ab @ ab)
    if denom < 1e-12:
        return np.linalg.norm(p - a, axis=1)
    t = np.clip(((p - a) @ ab) / denom, 0.0, 1.0)
    return np.linalg.norm(p - (a + t[:, None] * ab), axis=1)


# ----------------------------------------------------------------------- segments
def _boundary_loops(faces_sel: np.ndarray) -> Optional[List[List[int]]]:
    """Ordered boundary cycles of a face subset, following the faces' own edge
    direction.  Returns None if the boundary is not a set of simple cycles."""
    e = np.concatenate([faces_sel[:, [0, 1]], faces_sel[:, [1, 2]], faces_sel[:, [2, 0]]], 0)
    directed = set(map(tuple, e.tolist()))
    nxt: Dict[int, int] = {}
    for a, b in directed:
        if (b, a) not in directed:
            if a in nxt:
                return None  # pinch vertex
            nxt[a] = b
    loops = []
    seen = set()
    for start in sorted(nxt):
        if start in seen:
            continue
        loop = [start]
        seen.add(start)
        cur = nxt[start]
        while cur != start:
            if cur in seen or cur not in nxt:
                return None
            loop.append(cur)
            seen.add(cur)
            cur = nxt[cur]
        loops.append(loop)
    return loops


def _clean_face_selection(sel: np.ndarray, faces: np.ndarray, iters: int = 6) -> np.ndarray:
    """Remove ear faces / fill notches so the selection's boundary is smooth."""
    num_faces = len(faces)
    edges = np.sort(np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0), 1)
    key = edges[:, 0] * (faces.max() + 1) + edges[:, 1]
    order = np.argsort(key, kind='stable')
    fid = np.tile(np.arange(num_faces), 3)[order]
    # a closed manifold: every undirected edge appears exactly twice
    pair_a, pair_b = fid[0::2], fid[1::2]
    nbr = [[] for _ in range(num_faces)]
    for a, b in zip(pair_a.tolist(), pair_b.tolist()):
        nbr[a].append(b)
        nbr[b].append(a)
    nbr = np.asarray(nbr, dtype=np.int64)
    sel = sel.copy()
    for _ in range(iters):
        inside_nbrs = sel[nbr].sum(1)
        new = sel.copy()
        new[sel & (inside_nbrs <= 1)] = False
        new[~sel & (inside_nbrs >= 2)] = True
        if np.array_equal(new, sel):
            break
        sel = new
    return sel


def _remove_pinches(sel: np.ndarray, faces: np.ndarray, iters: int = 20) -> np.ndarray:
    """Drop selected faces around boundary vertices that have more than one
    outgoing boundary edge, until the boundary is a set of simple cycles."""
    sel = sel.copy()
    for _ in range(iters):
        f = faces[sel]
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
        directed = set(map(tuple, e.tolist()))
        out_deg: Dict[int, int] = {}
        for a, b in directed:
            if (b, a) not in directed:
                out_deg[a] = out_deg.get(a, 0) + 1
        pinch = [v for v, c in out_deg.items() if c > 1]
        if not pinch:
            break
        bad = np.isin(faces, np.asarray(pinch)).any(1)
        sel &= ~bad
    return sel


def _largest_component(sel: np.ndarray, faces: np.ndarray, num_verts: int) -> np.ndarray:
    idx = np.where(sel)[0]
    if len(idx) == 0:
        return sel
    f = faces[idx]
    rows = np.concatenate([idx, idx, idx])
    cols = np.concatenate([f[:, 0], f[:, 1], f[:, 2]])
    inc = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(len(faces), num_verts)).tocsr()
    g = (inc @ inc.T).tocsr()
    n, lab = sp.csgraph.connected_components(g[idx][:, idx], directed=False)
    best = np.argmax(np.bincount(lab))
    out = np.zeros_like(sel)
    out[idx[lab == best]] = True
    return out


def _segment_from_faces(fsel: np.ndarray, faces: np.ndarray, want_loops: int) -> Optional[dict]:
    if fsel.sum() < 4:
        return None
    loops = _boundary_loops(faces[fsel])
    if loops is None or len(loops) != want_loops:
        return None
    bands = {}
    for bi, loop in enumerate(loops):
        # repeat the first vertex so the reference's open fan
        # (segmentation.py:61-64: range(len-1)) closes the cap
        bands['band%d' % bi] = np.asarray(loop + [loop[0]], dtype=np.int64)
    return {'vidx': np.unique(faces[fsel]).astype(np.int64), 'bands': bands}


def _make_segments(verts, faces, limb, axial, rings: int, segs: int) -> Dict[str, dict]:
    specs = {
        'left_upperarm': (1, 0.28, 0.52), 'right_upperarm': (2, 0.28, 0.52),
        'left_forearm': (1, 0.58, 0.84), 'right_forearm': (2, 0.58, 0.84),
        'left_thigh': (3, 0.34, 0.60), 'right_thigh': (4, 0.34, 0.60),
    }
    axes = _limb_axes()
    out: Dict[str, dict] = {}
    # topological segments (exist at every resolution): vertex v>0 lies on UV ring (v-1)//segs
    ring_of = np.concatenate([[-1], np.repeat(np.arange(rings), segs), [rings]])
    fr = ring_of[faces]
    h, n0, n1 = max(rings // 6, 1), max(rings // 6, 1) + 1, max(rings // 3, 3)
    head = _segment_from_faces(fr.max(1) <= h, faces, 1)
    if head is not None:
        out['head'] = head
    neck = _segment_from_faces((fr.min(1) >= n0) & (fr.max(1) <= n1), faces, 2)
    if neck is not None:
        out['neckband'] = neck
    for name, (k, f0, f1) in specs.items():
        d, length, rad = axes[k]
        cen = verts[faces].mean(1)
        cen_limb = np.stack([_point_segment_distance(cen, np.zeros(3), dd * ll) - rr
                             for dd, ll, rr in axes], 1).argmin(1)
        cen_ax = cen @ d
        fsel = (cen_limb == k) & (cen_ax >= f0 * length) & (cen_ax <= f1 * length)
        fsel = _clean_face_selection(fsel, faces)
        fsel = _largest_component(fsel, faces, len(verts))
        fsel = _remove_pinches(fsel, faces)
        fsel = _largest_component(fsel, faces, len(verts))
        seg = _segment_from_faces(fsel, faces, 2) if fsel.sum() >= 8 else None
        if seg is not None:
            out[name] = seg
    return out


def _painted_segment(paint: np.ndarray, faces: np.ndarray, num_verts: int, max_loops: int,
                     strays: int, rng) -> Optional[dict]:
    """A segment as the reference defines one (tuch/utils/segmentation.py:40-55): a painted VERTEX set; its faces are the
    body faces with all three corners painted, its bands the ordered boundary loops.  ``paint`` is made consistent
    (pinch vertices unpainted, everything outside the largest face component unpainted) so that the boundary is a set
    of simple cycles -- however many: where two limbs merge a "band" is an open patch with ONE loop --; then ``strays`` painted vertices that belong to no segment face are added (a painted .ply may
    hold such vertices; the reference tests them against the segment mesh like any other)."""
    paint = paint.copy()
    for _ in range(30):
        fsel = paint[faces].all(1)
        if fsel.sum() < 8:
            return None
        big = _largest_component(fsel, faces, num_verts)
        keep = np.zeros(num_verts, bool)
        keep[np.unique(faces[big])] = True
        f = faces[keep[faces].all(1)]
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
        directed = set(map(tuple, e.tolist()))
        out_deg: Dict[int, int] = {}
        for a, b in directed:
            if (b, a) not in directed:
                out_deg[a] = out_deg.get(a, 0) + 1
        pinch = [v for v, c in out_deg.items() if c > 1]
        if not pinch and np.array_equal(keep, paint):
            break
        keep[pinch] = False
        paint = keep
    else:
        return None
    fsel = paint[faces].all(1)
    loops = _boundary_loops(faces[fsel])
    if loops is None or not 1 <= len(loops) <= max_loops:
        return None
    # stray painted vertices: not next to any painted vertex, so they complete no face
    adj = _adjacency(num_verts, faces)
    near = (adj @ (adj @ paint.astype(np.float64))) > 0
    ring2 = (adj @ near.astype(np.float64)) > 0
    cand = np.where(ring2 & ~near & ~paint)[0]
    vidx = np.where(paint)[0]
    if strays and len(cand):
        pick = []
        for v in rng.permutation(cand).tolist():
            if all(abs(v - q) > 0 and adj[v, q] == 0 for q in pick):
                pick.append(v)
            if len(pick) == strays:
                break
        vidx = np.sort(np.concatenate([vidx, np.asarray(pick, np.int64)]))
    bands = {'band%d' % bi: np.asarray(loop + [loop[0]], dtype=np.int64) for bi, loop in enumerate(loops)}
    return {'vidx': vidx.astype(np.int64), 'bands': bands}


def _make_painted_segments(verts, faces, rng) -> Dict[str, dict]:
    """Segments of the irregular-topology body: vertices painted by a noisy geometric rule (ragged boundaries that
    follow no edge loop of the mesh), eight of them like the UV body's."""
    specs = {
        'head': (0, 0.78, 9.0, 1), 'neckband': (0, 0.55, 0.74, 2),
        'left_upperarm': (1, 0.25, 0.55, 2), 'right_upperarm': (2, 0.25, 0.55, 2),
        'left_forearm': (1, 0.58, 0.90, 2), 'right_forearm': (2, 0.58, 0.90, 2),
        'left_thigh': (3, 0.30, 0.64, 2), 'right_thigh': (4, 0.30, 0.64, 2),
    }
    axes = _limb_axes()
    num_verts = len(verts)
    v_limb = np.stack([_point_segment_distance(verts, np.zeros(3), dd * ll) - rr for dd, ll, rr in axes], 1).argmin(1)
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0)
    elen = np.linalg.norm(verts[e[:, 0]] - verts[e[:, 1]], axis=1)
    edge_len = np.bincount(e[:, 0], elen, num_verts) / np.maximum(np.bincount(e[:, 0], minlength=num_verts), 1)
    out: Dict[str, dict] = {}
    for si, (name, (k, f0, f1, loops)) in enumerate(specs.items()):
        d, length, _ = axes[k]
        ax = verts @ d + 0.2 * edge_len * rng.standard_normal(num_verts)    # ragged: per-vertex jitter of the cut
        paint = (v_limb == k) & (ax >= f0 * length) & (ax <= f1 * length)
        seg = _painted_segment(paint, faces, num_verts, 4, strays=2 if si % 2 == 0 else 0, rng=rng)
        if seg is not None:
            out[name] = seg
    return out


# -------------------------------------------------------------------------- main
def make_body(rings: int = 84, segs: int = 82, seed: int = 1234, with_geodesics: bool = True,
              relax_iters: int = 200, num_regions: int = 24, hd_samples_per_face: int = 3,
              geothres_for_pairs: float = 0.3, topology: str = 'uv', freq: int = 26,
              flips: Optional[int] = None) -> SyntheticBody:
    """topology 'uv': lat-long sphere of ``rings`` x ``segs`` (valence 6 + two poles, V = 6890 at (84, 82)).
    topology 'ico': geodesic icosahedron of frequency ``freq`` (V = 10 freq^2 + 2: 6762 at 26) with ``flips`` random
    edge flips (default V / 10) -> valences 4 ... 9, painted segments with ragged boundaries and stray vertices; what
    the kernels meet on a mesh that is not a regular grid (the reference loads the SMPL mesh as it is,
    tuch/models/smpl.py:37-42, tuch/utils/segmentation.py:40-66)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    if topology == 'ico':
        topo_rng = np.random.Generator(np.random.PCG64(seed + 77))
        dirs, faces = _ico_topology(freq)
        # a random rotation: no vertex on a symmetry plane of the limbs
        q, _ = np.linalg.qr(topo_rng.standard_normal((3, 3)))
        dirs = dirs @ (q * np.sign(np.linalg.det(q))).T
        faces = _random_edge_flips(dirs, faces, len(dirs) // 10 if flips is None else flips, topo_rng)
    elif topology == 'uv':
        dirs, faces = _uv_sphere_topology(rings, segs)
    else:
        raise ValueError('topology must be uv or ico, got %r' % (topology,))
    num_verts = dirs.shape[0]
    adj = _adjacency(num_verts, faces)
    dirs = _relax(dirs, faces, relax_iters)
    verts = dirs * _radius(dirs)[:, None]
    faces = _orient_outward(verts, faces)

    # limb assignment and axial coordinate
    axes = _limb_axes()
    limb_d = np.stack([_point_segment_distance(verts, np.zeros(3), d * L) - r for d, L, r in axes], 1)
    limb = limb_d.argmin(1)
    axial = np.stack([verts @ d for d, _, _ in axes], 1)[np.arange(num_verts), limb]

    # skeleton and skinning
    joints = _joint_positions()
    first_child = {j: j for j in range(24)}
    for c in range(23, 0, -1):
        first_child[int(SMPL_PARENTS[c])] = c
    sigma = 0.08
    w = np.zeros((num_verts, 24))
    for j in range(24):
        dist = _point_segment_distance(verts, joints[j], joints[first_child[j]])
        w[:, j] = np.exp(-0.5 * (dist / sigma) ** 2)
    keep = np.argsort(-w, axis=1)[:, :4]
    mask = np.zeros_like(w, dtype=bool)
    np.put_along_axis(mask, keep, True, axis=1)
    w = np.where(mask, w, 0.0) + 1e-12 * mask
    w /= w.sum(1, keepdims=True)

    jr = np.exp(-0.5 * (np.linalg.norm(verts[None] - joints[:, None], axis=2) / 0.06) ** 2) + 1e-9
    jr /= jr.sum(1, keepdims=True)

    shapedirs = _smooth(rng.standard_normal((num_verts, 30)), adj, 12)
    shapedirs = (0.01 * shapedirs / shapedirs.std()).reshape(num_verts, 3, 10)
    posedirs = _smooth(rng.standard_normal((num_verts, 3 * 207)), adj, 8)
    posedirs = 0.002 * posedirs / posedirs.std()
    # [V, 3, 207] -> [207, V*3] (smplx stores posedirs as [P, V*3])
    posedirs = posedirs.reshape(num_verts, 3, 207).transpose(2, 0, 1).reshape(207, num_verts * 3)

    anchors = rng.standard_normal((256, 3))
    anchors /= np.linalg.norm(anchors, axis=1, keepdims=True)
    anchor_pts = anchors * _radius(anchors)[:, None]
    nearest = np.linalg.norm(verts[None] - anchor_pts[:, None], axis=2).argmin(1)
    extra_vertex_ids = np.asarray(list(dict.fromkeys(nearest.tolist()))[:21], dtype=np.int64)
    assert len(extra_vertex_ids) == 21
    jx_centres = anchor_pts[-9:]
    jrx = np.exp(-0.5 * (np.linalg.norm(verts[None] - jx_centres[:, None], axis=2) / 0.05) ** 2) + 1e-9
    jrx /= jrx.sum(1, keepdims=True)
    joint_map = rng.permutation(54)[:49].astype(np.int64)

    # geodesics (graph shortest paths on template edges)
    geod = None
    if with_geodesics:
        coo = adj.tocoo()
        wgt = np.linalg.norm(verts[coo.row] - verts[coo.col], axis=1)
        graph = sp.csr_matrix((wgt, (coo.row, coo.col)), shape=adj.shape)
        geod = dijkstra(graph, directed=False).astype(np.float32)

    # contact regions: farthest-point seeds + nearest-seed cells (Euclidean on the
    # template is enough for a partition; pairs are filtered by geodesic distance)
    seeds = [int(np.argmax(verts[:, 1]))]
    dmin = np.linalg.norm(verts - verts[seeds[0]], axis=1)
    for _ in range(num_regions - 1):
        seeds.append(int(np.argmax(dmin)))
        dmin = np.minimum(dmin, np.linalg.norm(verts - verts[seeds[-1]], axis=1))
    cell = np.linalg.norm(verts[:, None] - verts[seeds][None], axis=2).argmin(1)
    names = ['reg%02d' % i for i in range(num_regions)]
    regions = {names[i]: np.where(cell == i)[0].astype(np.int64) for i in range(num_regions)}
    pairs = []
    for i in range(num_regions):
        for k in range(i + 1, num_regions):
            if geod is not None:
                far = geod[seeds[i], seeds[k]] > 2.0 * geothres_for_pairs
            else:
                far = np.linalg.norm(verts[seeds[i]] - verts[seeds[k]]) > 2.0 * geothres_for_pairs
            if far:
                pairs.append((names[i], names[k]))

    if topology == 'ico':
        segments = _make_painted_segments(verts, faces, topo_rng)
    else:
        segments = _make_segments(verts, faces, limb, axial, rings, segs)

    # HD regressor: fixed barycentric samples on every face
    bary = np.array([[0.6, 0.2, 0.2], [0.2, 0.6, 0.2], [0.2, 0.2, 0.6], [1 / 3, 1 / 3, 1 / 3]])
    bary = bary[:hd_samples_per_face]
    hd_face_id = np.repeat(np.arange(len(faces)), len(bary)).astype(np.int64)
    hd_idx = faces[hd_face_id]
    hd_w = np.tile(bary, (len(faces), 1)).astype(np.float32)

    # GMM prior (8 x 69-D): random SPD covariances, small means
    means = 0.2 * rng.standard_normal((8, 69))
    covs = []
    for _ in range(8):
        a = rng.standard_normal((69, 69)) * 0.05
        covs.append(a @ a.T + 0.05 * np.eye(69))
    weights = rng.random(8) + 0.2
    weights /= weights.sum()
    gmm = {'means': means, 'covars': np.stack(covs), 'weights': weights}

    return SyntheticBody(
        v_template=verts.astype(np.float32), faces=faces,
        shapedirs=shapedirs.astype(np.float32), posedirs=posedirs.astype(np.float32),
        J_regressor=jr.astype(np.float32), lbs_weights=w.astype(np.float32),
        parents=SMPL_PARENTS.copy(), extra_vertex_ids=extra_vertex_ids,
        J_regressor_extra=jrx.astype(np.float32), joint_map=joint_map,
        geodesics=geod, regions=regions, region_pairs=pairs, segments=segments,
        hd_regressor=np.zeros((0, 0), np.float32), hd_bary_idx=hd_idx, hd_bary_w=hd_w,
        hd_face_id=hd_face_id, gmm=gmm,
        limb_of_vertex=limb.astype(np.int64), axial_of_vertex=axial.astype(np.float32))


def dense_hd_regressor(body: SyntheticBody) -> np.ndarray:
    """Dense [N_hd, V] matrix as the reference stores it (loss.py:81-83). Small meshes only."""
    n = len(body.hd_face_id)
    out = np.zeros((n, body.num_verts), np.float32)
    rows = np.repeat(np.arange(n), 3)
    np.add.at(out, (rows, body.hd_bary_idx.ravel()), body.hd_bary_w.ravel())
    return out


def random_poses(batch: int, seed: int, penetrating_fraction: float = 0.5):
    """Pose/shape batch of SURVEY.md §8d: body_pose ~ 0.25 N(0,1) clipped, half the
    bodies get an arm-across-torso bias so that self-penetration occurs."""
    rng = np.random.Generator(np.random.PCG64(seed))
    body_pose = np.clip(0.25 * rng.standard_normal((batch, 69)), -1.2, 1.2)
    global_orient = 0.1 * rng.standard_normal((batch, 3))
    betas = np.clip(rng.standard_normal((batch, 10)), -2.0, 2.0)
    n_pen = int(round(batch * penetrating_fraction))
    for b in range(n_pen):
        # swing the left arm (joints 13,16 -> body_pose rows 12,15) down across the torso
        # and the right arm toward the head; amounts vary per body
        amt = 0.8 + 0.5 * rng.random()
        body_pose[b, 3 * 15 + 2] -= 1.1 * amt      # L shoulder about z: arm down to the trunk
        body_pose[b, 3 * 17 + 1] += 0.6 * amt      # L elbow
        body_pose[b, 3 * 16 + 2] += 1.0 * amt      # R shoulder about z
        body_pose[b, 3 * 0 + 2] += 0.35 * amt      # L hip about z: legs together
        body_pose[b, 3 * 1 + 2] -= 0.35 * amt
    return (body_pose.astype(np.float32), global_orient.astype(np.float32), betas.astype(np.float32))


def through_pose(batch: int, seed: int):
    """Poses for parity cases where a limb goes THROUGH the body instead of resting against it: body 0 has the left
    upper arm down along the trunk and the elbow bent inward so that the forearm passes through the trunk / hip
    (about half of the forearm's vertices end up inside); further bodies are ordinary ``random_poses``."""
    body_pose, global_orient, betas = random_poses(batch, seed, penetrating_fraction=0.0)
    body_pose[0] = 0.0
    body_pose[0, 3 * 15 + 2] = -1.2      # L shoulder about z
    body_pose[0, 3 * 17 + 2] = -1.5      # L elbow about z: forearm across the body
    global_orient[0] = 0.0
    betas[0] = 0.0
    return body_pose, global_orient, betas


def folded_poses(batch: int, seed: int):
    """Heavily self-penetrating poses (measurement of the worst case, not anatomy): on top of ``random_poses`` with
    every body penetrating, a third of the bodies get a forearm through the trunk, a third the legs crossed through
    each other, a third the trunk folded forward over the thighs."""
    rng = np.random.Generator(np.random.PCG64(seed + 5))
    body_pose, global_orient, betas = random_poses(batch, seed, penetrating_fraction=1.0)
    for b in range(batch):
        amt = 0.8 + 0.4 * rng.random()
        if b % 3 == 0:
            body_pose[b, 3 * 15 + 2] = -1.2 * amt
            body_pose[b, 3 * 17 + 2] = -1.5 * amt
        elif b % 3 == 1:
            body_pose[b, 3 * 0 + 2] += 0.9 * amt       # hips about z: legs cross
            body_pose[b, 3 * 1 + 2] -= 0.9 * amt
        else:
            body_pose[b, 3 * 2 + 0] += 0.9 * amt       # spine joints about x: fold forward
            body_pose[b, 3 * 5 + 0] += 0.9 * amt
            body_pose[b, 3 * 0 + 0] -= 0.8 * amt       # thighs up
            body_pose[b, 3 * 1 + 0] -= 0.8 * amt
    return body_pose.astype(np.float32), global_orient, betas


# ------------------------------------------------------------------ asset files in the reference's formats
def _write_ply_vertex_colours(path: str, verts: np.ndarray, red: np.ndarray, binary: bool = True) -> None:
    import struct
    header = ['ply', 'format %s 1.0' % ('binary_little_endian' if binary else 'ascii'),
              'element vertex %d' % len(verts), 'property float x', 'property float y', 'property float z',
              'property uchar red', 'property uchar green', 'property uchar blue', 'property uchar alpha',
              'element face 0', 'property list uchar int vertex_indices', 'end_header']
    with open(path, 'wb') as f:
        f.write(('\n'.join(header) + '\n').encode())
        for v, r in zip(verts, red):
            if binary:
                f.write(struct.pack('<fffBBBB', v[0], v[1], v[2], int(r), 0, 0, 255))
            else:
                f.write(('%f %f %f %d 0 0 255\n' % (v[0], v[1], v[2], int(r))).encode())


def write_reference_assets(body: SyntheticBody, root: str, with_extra_vertex_ids: Optional[bool] = None) -> Dict[str, str]:
    """Write ``body`` as the files the reference loads, under ``root`` with the reference's relative paths
    (configs/config.py:74-92), plus a ``configs/config.py`` and the two python modules of the data folder
    (``data/essentials/constants.py``, ``data/essentials/segments/smpl/segm_utils.py``).  With ``root`` as
    the working directory and on sys.path, the reference's scripts' constructor calls (train.py:57-100,
    demo_smplify_dc.py:54-87) run against synthetic data.

    The SMPL pickle has the official key names and layouts (posedirs [V,3,207], sparse J_regressor,
    kintree_table [2,24], f).  A mesh that does not have the 6890-vertex SMPL topology carries the extra key
    ``extra_vertex_ids`` (the 21 picked vertices smplx takes from its own table), see models/smpl.py.
    Returns the written paths by config name."""
    import os
    import pickle
    from tuch_amd.models.smpl import SPIN_JOINT_NAMES
    if with_extra_vertex_ids is None:
        with_extra_vertex_ids = body.num_verts != 6890
    j = lambda *p: os.path.join(root, *p)
    paths = {'SMPL_MODEL_DIR': 'data/models/smpl', 'JOINT_REGRESSOR_TRAIN_EXTRA': 'data/essentials/spin/J_regressor_extra.npy',
             'PRIOR_FOLDER': 'data/essentials/spin', 'GEODESICS_SMPL': 'data/essentials/geodesics/smpl/smpl_neutral_geodesic_dist.npy',
             'HD_MODEL_DIR': 'data/essentials/hd_model/smpl', 'SEGMENT_DIR': 'data/essentials/segments/smpl',
             'STATIC_FITS_DIR': 'data/static_fits', 'DSC_ROOT': 'data/dsc'}
    for d in ('SMPL_MODEL_DIR', 'PRIOR_FOLDER', 'HD_MODEL_DIR', 'SEGMENT_DIR', 'STATIC_FITS_DIR', 'DSC_ROOT'):
        os.makedirs(j(paths[d]), exist_ok=True)
    os.makedirs(j(os.path.dirname(paths['GEODESICS_SMPL'])), exist_ok=True)
    os.makedirs(j('configs'), exist_ok=True)
    # SMPL model, official layout
    v = body.num_verts
    kintree = np.stack([np.where(body.parents < 0, 2 ** 32 - 1, body.parents), np.arange(24)]).astype(np.int64)
    model = {'v_template': body.v_template.astype(np.float64), 'shapedirs': body.shapedirs.astype(np.float64),
             'posedirs': body.posedirs.reshape(207, v, 3).transpose(1, 2, 0).astype(np.float64),
             'J_regressor': sp.csc_matrix(body.J_regressor.astype(np.float64)), 'weights': body.lbs_weights.astype(np.float64),
             'kintree_table': kintree, 'f': body.faces.astype(np.uint32)}
    if with_extra_vertex_ids:
        model['extra_vertex_ids'] = body.extra_vertex_ids
    with open(j(paths['SMPL_MODEL_DIR'], 'SMPL_NEUTRAL.pkl'), 'wb') as f:
        pickle.dump(model, f, protocol=2)
    np.save(j(paths['JOINT_REGRESSOR_TRAIN_EXTRA']), body.J_regressor_extra)
    with open(j(paths['PRIOR_FOLDER'], 'gmm_08.pkl'), 'wb') as f:
        pickle.dump({k: np.asarray(val, np.float64) for k, val in body.gmm.items()}, f, protocol=2)
    if body.geodesics is not None:
        np.save(j(paths['GEODESICS_SMPL']), body.geodesics)
    # HD regressor: the reference stores it dense [N_hd, V] float (1.1 GB at SMPL size)
    np.save(j(paths['HD_MODEL_DIR'], 'smpl_neutral_hd_vert_regressor.npy'), dense_hd_regressor(body))
    with open(j(paths['HD_MODEL_DIR'], 'smpl_neutral_hd_sample_from_mesh_out.pkl'), 'wb') as f:
        pickle.dump({'faces_vert_is_sampled_from': body.hd_face_id}, f, protocol=2)
    # segments: painted .ply + segm_utils.py
    for i, (name, seg) in enumerate(body.segments.items()):
        red = np.zeros(v, np.int64)
        red[seg['vidx']] = 255
        _write_ply_vertex_colours(j(paths['SEGMENT_DIR'], 'smpl_segment_%s.ply' % name), body.v_template, red, binary=bool(i % 2))
    with open(j(paths['SEGMENT_DIR'], 'segm_utils.py'), 'w') as f:
        f.write('# synthetic stand-in for the licensed segm_utils.py: ordered boundary loops per segment\nsegments = {\n')
        for name, seg in body.segments.items():
            f.write('    %r: {\n' % name)
            for band, loop in seg['bands'].items():
                f.write('        %r: %r,\n' % (band, [int(x) for x in loop]))
            f.write('    },\n')
        f.write('}\n')
    # DSC region tables
    with open(j(paths['DSC_ROOT'], 'classes.pkl'), 'wb') as f:
        pickle.dump(np.asarray(body.region_pairs), f, protocol=2)
    with open(j(paths['DSC_ROOT'], 'ContactSigSMPL.pkl'), 'wb') as f:
        pickle.dump({k: [int(x) for x in val] for k, val in body.regions.items()}, f, protocol=2)
    # constants.py: SPIN's joint names, the joint map of THIS body, flip permutation
    flip = [0, 2, 1, 3, 5, 4, 6, 8, 7, 9, 11, 10, 12, 14, 13, 15, 17, 16, 19, 18, 21, 20, 23, 22]
    with open(j('data/essentials/constants.py'), 'w') as f:
        f.write('# synthetic stand-in for the licensed constants.py\nFOCAL_LENGTH = 5000.\nIMG_RES = 224\n')
        f.write('JOINT_NAMES = %r\n' % (list(SPIN_JOINT_NAMES),))
        f.write('JOINT_IDS = {JOINT_NAMES[i]: i for i in range(len(JOINT_NAMES))}\n')
        f.write('JOINT_MAP = %r\n' % ({n: int(body.joint_map[i]) for i, n in enumerate(SPIN_JOINT_NAMES)},))
        f.write('SMPL_JOINTS_FLIP_PERM = %r\n' % (flip,))
        f.write('SMPL_POSE_FLIP_PERM = [3 * i + k for i in SMPL_JOINTS_FLIP_PERM for k in range(3)]\n')
    with open(j('configs/config.py'), 'w') as f:
        f.write('# synthetic stand-in for configs/config.py of the reference (same names)\n')
        for k, val in paths.items():
            f.write('%s = %r\n' % (k, val))
        f.write('geothres = 0.3\neuclthres = 0.02\n')
    return paths


# ------------------------------------------------------------------ stand-ins for the caller's side of a training step
def make_regressor(seed: int, img: int = 8):
    """A deterministic stand-in for the HMR regressor (``hmr(...)`` in train.py): images [B,3,img,img] ->
    (pred_rotmat [B,24,3,3], pred_betas [B,10], pred_camera [B,3]).  One linear layer (weights from NumPy PCG64, so the
    same on every device and torch version) followed by the 6D -> rotation-matrix map; the outputs are small
    perturbations of a mean pose, which is what a trained regressor hands to the losses."""
    import torch
    import torch.nn as nn

    class Regressor(nn.Module):
        def __init__(self):
            super().__init__()
            rng = np.random.Generator(np.random.PCG64(seed))
            self.fc = nn.Linear(3 * img * img, 24 * 6 + 13)
            with torch.no_grad():
                self.fc.weight.copy_(torch.tensor(0.02 * rng.standard_normal((24 * 6 + 13, 3 * img * img)), dtype=torch.float32))
                self.fc.bias.copy_(torch.tensor(0.05 * rng.standard_normal(24 * 6 + 13), dtype=torch.float32))
            base = np.tile(np.array([1.0, 0.0, 0.0, 1.0, 0.0, 0.0], np.float32), 24)       # identity in 6D form
            self.register_buffer('mean6d', torch.tensor(base))

        def forward(self, images):
            x = self.fc(images.flatten(1))
            r6 = (self.mean6d + x[:, :144]).view(-1, 3, 2)
            a1, a2 = r6[:, :, 0], r6[:, :, 1]
            b1 = torch.nn.functional.normalize(a1)
            b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1)
            rotmat = torch.stack((b1, b2, torch.cross(b1, b2, dim=1)), dim=-1).view(-1, 24, 3, 3)
            betas = x[:, 144:154]
            camera = torch.stack([0.9 + 0.1 * torch.tanh(x[:, 154]), 0.1 * torch.tanh(x[:, 155]), 0.1 * torch.tanh(x[:, 156])], 1)
            return rotmat, betas, camera
    return Regressor()


def make_train_batch(body: SyntheticBody, batch: int, seed: int, datasets=(('dsA', 50), ('dsB', 30)), img: int = 8) -> dict:
    """One ``input_batch`` of the reference's training step (train_module.py:105-134) as NumPy arrays + lists."""
    rng = np.random.Generator(np.random.PCG64(seed))
    bp, go, be = random_poses(batch, seed)
    names = [datasets[int(i)][0] for i in rng.integers(0, len(datasets), batch)]
    sizes = dict(datasets)
    kp = np.concatenate([rng.uniform(-0.8, 0.8, (batch, 49, 2)), rng.uniform(0.3, 1.0, (batch, 49, 1))], 2).astype(np.float32)
    has_smpl = rng.random(batch) < 0.25
    return {
        'img': rng.standard_normal((batch, 3, img, img)).astype(np.float32),
        'sample_index': np.asarray([rng.integers(0, sizes[n]) for n in names], np.int64),
        'is_flipped': (rng.random(batch) < 0.5).astype(np.int64),
        'rot_angle': (rng.uniform(-30, 30, batch) * (rng.random(batch) < 0.6)).astype(np.float32),
        'dataset_name': names,
        'has_pose_3d': (rng.random(batch) < 0.5).astype(np.uint8),
        'has_disc_contact': (rng.random(batch) < 0.6).astype(np.uint8),
        'has_gt_kpts': (rng.random(batch) < 0.4).astype(np.uint8),
        'has_smpl': has_smpl.astype(np.uint8),
        'has_pgt_smpl': (rng.random(batch) < 0.1).astype(np.uint8),
        'keypoints': kp,
        'pose_3d': np.concatenate([0.3 * rng.standard_normal((batch, 24, 3)), rng.uniform(0.5, 1.0, (batch, 24, 1))], 2).astype(np.float32),
        'pose': np.concatenate([go, bp], 1).astype(np.float32),
        'betas': be,
        'contact_vec': (rng.random((batch, len(body.region_pairs))) < 0.05).astype(np.float32),
    }
