"""CPU oracle for the self-contact path -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

NumPy float32 composition over the C primitives in tuch_oracle.c.  Every function
cites the reference lines it restates (paths relative to the reference repo root).
Gradients are written out analytically and pinned against
the reference's autograd through tests/golden/*.npz.

The bmm-form squared distance (contact.py:27-42) cannot be reproduced bit-for-bit
(BLAS summation order); values derived from it are compared with an absolute
tolerance of a few 1e-7 * |x|^2 in the tests, see DESIGN.md "Parity".
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags='C_CONTIGUOUS')
_f64p = np.ctypeslib.ndpointer(np.float64, flags='C_CONTIGUOUS')
_i64p = np.ctypeslib.ndpointer(np.int64, flags='C_CONTIGUOUS')
_u8p = np.ctypeslib.ndpointer(np.uint8, flags='C_CONTIGUOUS')


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, 'libtuch_oracle.so')
    src = os.path.join(_HERE, 'tuch_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(['make', '-C', _HERE, '-B', 'libtuch_oracle.so'], check=True,
                       stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.oracle_set_threads.argtypes = [ctypes.c_int]
        L.oracle_max_threads.restype = ctypes.c_int
        L.oracle_pairwise_sq.argtypes = [_f32p, ctypes.c_int, _f32p, ctypes.c_int, _f32p]
        L.oracle_v2v_min_masked.argtypes = [_f32p, ctypes.c_int, _u8p, _f32p, _i64p]
        L.oracle_solid_angles.argtypes = [_f32p, ctypes.c_int, _f32p, ctypes.c_int, _f32p]
        L.oracle_winding.argtypes = [_f32p, ctypes.c_int, _f32p, ctypes.c_int, _f32p]
        L.oracle_gather_tris.argtypes = [_f32p, _i64p, ctypes.c_int, _f32p]
        L.oracle_region_min_f64.argtypes = [_f32p, _i64p, ctypes.c_int, _i64p, ctypes.c_int,
                                            _f64p, _i64p, _i64p]
        _LIB = L
    return _LIB


def set_threads(n: int) -> None:
    lib().oracle_set_threads(int(n))


def max_threads() -> int:
    return int(lib().oracle_max_threads())


def _c32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ci64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int64)


# ------------------------------------------------------------------ primitives
def pairwise_sq(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """tuch/utils/contact.py:23-47 for one batch element: [Nx,3],[Ny,3] -> [Nx,Ny]."""
    x, y = _c32(x), _c32(y)
    out = np.empty((x.shape[0], y.shape[0]), np.float32)
    lib().oracle_pairwise_sq(x, x.shape[0], y, y.shape[0], out)
    return out


def pairwise_adjoint(x: np.ndarray, y: np.ndarray, g: np.ndarray, squared: bool = True):
    """What torch autograd returns through tuch/utils/contact.py:23-47 for one batch element: P = rx + ry - 2 x y^T with
    rx, ry the diagonals of x x^T and y y^T (:27-43), so dL/dx_i = 2 sum_j g_ij (x_i - y_j), dL/dy_j = 2 sum_i g_ij
    (y_j - x_i); sqrt (:45) contributes g / (2 sqrt(P)).  float64 accumulation, float32 result."""
    x64, y64, g64 = np.asarray(x, np.float64), np.asarray(y, np.float64), np.asarray(g, np.float64)
    if not squared:
        g64 = g64 / (2.0 * np.sqrt(pairwise_sq(x, y).astype(np.float64)))
    gx = 2.0 * (x64 * g64.sum(1)[:, None] - g64 @ y64)
    gy = 2.0 * (y64 * g64.sum(0)[:, None] - g64.T @ x64)
    return gx.astype(np.float32), gy.astype(np.float32)


def solid_angle_adjoint(points: np.ndarray, tris: np.ndarray, g: np.ndarray):
    """What torch autograd returns through tuch/utils/contact.py:79-109 for one batch element: points [Q,3], tris [F,3,3],
    g [Q,F] = dL/dOmega -> (dL/dpoints [Q,3], dL/dtris [F,3,3]).  Omega = 2 atan2(num, den), num = a.(b x c),
    den = |a||b||c| + (a.b)|c| + (a.c)|b| + (b.c)|a| with a, b, c = corners - point (:80-100).  float64."""
    p = np.asarray(points, np.float64)[:, None, None, :]
    t = np.asarray(tris, np.float64)[None]
    c = t - p                                            # [Q,F,3,3]
    a, b, cc = c[:, :, 0], c[:, :, 1], c[:, :, 2]
    la, lb, lc = (np.linalg.norm(x, axis=-1) for x in (a, b, cc))
    bc, ca, ab = np.cross(b, cc), np.cross(cc, a), np.cross(a, b)
    num = (a * bc).sum(-1)
    dab, dac, dbc = (a * b).sum(-1), (a * cc).sum(-1), (b * cc).sum(-1)
    den = la * lb * lc + dab * lc + dac * lb + dbc * la
    k = 2.0 * np.asarray(g, np.float64) / (num * num + den * den)
    unit = lambda x, l: x / np.where(l > 0, l, 1.0)[..., None] * (l > 0)[..., None]
    da = unit(a, la) * (lb * lc + dbc)[..., None] + b * lc[..., None] + cc * lb[..., None]
    db = unit(b, lb) * (la * lc + dac)[..., None] + a * lc[..., None] + cc * la[..., None]
    dc = unit(cc, lc) * (la * lb + dab)[..., None] + a * lb[..., None] + b * la[..., None]
    ga = k[..., None] * (den[..., None] * bc - num[..., None] * da)
    gb = k[..., None] * (den[..., None] * ca - num[..., None] * db)
    gc = k[..., None] * (den[..., None] * ab - num[..., None] * dc)
    gt = np.stack([ga.sum(0), gb.sum(0), gc.sum(0)], 1)
    gp = -(ga + gb + gc).sum(1)
    return gp.astype(np.float32), gt.astype(np.float32)


def v2v_min_masked(verts: np.ndarray, geomask: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """tuch/smplify/losses.py:92-93, tuch/train/loss.py:269-270 -> (min_d2[V], argmin[V])."""
    verts = _c32(verts)
    gm = np.ascontiguousarray(geomask, dtype=np.uint8)
    v = verts.shape[0]
    mn = np.empty(v, np.float32)
    arg = np.empty(v, np.int64)
    lib().oracle_v2v_min_masked(verts, v, gm, mn, arg)
    return mn, arg


def gather_tris(verts: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """verts[faces] -> [F,3,3] (losses.py:81, loss.py:260, segmentation.py:77)."""
    verts, faces = _c32(verts), _ci64(faces)
    out = np.empty((faces.shape[0], 3, 3), np.float32)
    lib().oracle_gather_tris(verts, faces, faces.shape[0], out)
    return out


def solid_angles(points: np.ndarray, tris: np.ndarray) -> np.ndarray:
    """tuch/utils/contact.py:49-109 -> [Q,F]."""
    points, tris = _c32(points), _c32(tris)
    out = np.empty((points.shape[0], tris.shape[0]), np.float32)
    lib().oracle_solid_angles(points, points.shape[0], tris, tris.shape[0], out)
    return out


def winding_numbers(points: np.ndarray, tris: np.ndarray) -> np.ndarray:
    """tuch/utils/contact.py:112-147 -> [Q]."""
    points, tris = _c32(points), _c32(tris)
    out = np.empty(points.shape[0], np.float32)
    lib().oracle_winding(points, points.shape[0], tris, tris.shape[0], out)
    return out


def region_min_f64(verts, r1, r2):
    mn = np.zeros(1, np.float64)
    i = np.zeros(1, np.int64)
    j = np.zeros(1, np.int64)
    r1, r2 = _ci64(r1), _ci64(r2)
    lib().oracle_region_min_f64(_c32(verts), r1, len(r1), r2, len(r2), mn, i, j)
    return float(mn[0]), int(i[0]), int(j[0])


# -------------------------------------------------------------------- segments
class Segment:
    """tuch/utils/segmentation.py:29-99 restated on plain arrays.

    vidx   -- vertices of the segment (the .ply's red vertices, :42)
    bands  -- ordered boundary loops (segm_utils.segments[name] values, :45-46)
    faces  -- body faces fully inside vidx (:51-52) followed by the cap fans
              [b[i+1], b[i], V+k] for i in range(len(b)-1) (:56-66); the cap
              vertex of band k has index (faces.max()) + 1 + k (:37,:62)
    """

    def __init__(self, name: str, body_faces: np.ndarray, vidx: np.ndarray,
                 bands: Sequence[np.ndarray]):
        self.name = name
        body_faces = np.asarray(body_faces, np.int64)
        self.vidx = np.asarray(vidx, np.int64)
        self.bands = [np.asarray(b, np.int64) for b in bands]
        append_idx = int(body_faces.max())
        inside = np.isin(body_faces, self.vidx).sum(1) == 3
        caps = []
        for k, b in enumerate(self.bands):
            new = append_idx + 1 + k
            caps += [[b[i + 1], b[i], new] for i in range(len(b) - 1)]
        self.faces = np.concatenate([body_faces[inside], np.asarray(caps, np.int64).reshape(-1, 3)], 0)

    def closed_tris(self, verts: np.ndarray) -> np.ndarray:
        """segmentation.py:68-79: append one mean vertex per band, gather triangles."""
        verts = _c32(verts)
        caps = [verts[b].mean(0, dtype=np.float32) for b in self.bands]
        ext = np.concatenate([verts, np.asarray(caps, np.float32).reshape(-1, 3)], 0)
        return gather_tris(ext, self.faces)

    def exterior(self, verts: np.ndarray) -> np.ndarray:
        """segmentation.py:81-99: winding of the segment's verts vs its closed mesh, .le(0.99)."""
        w = winding_numbers(_c32(verts)[self.vidx], self.closed_tris(verts))
        return w <= np.float32(0.99)


def exterior_flags(verts: np.ndarray, faces: np.ndarray, segments: Optional[List[Segment]],
                   always_filter: bool) -> Tuple[np.ndarray, np.ndarray]:
    """Exterior flags for one body.

    losses.py:81-89 (SMPLify: the segment filter runs only if some vertex is
    interior) / loss.py:260-266 (train: always).  Returns (exterior[V] bool, w[V]).
    """
    verts = _c32(verts)
    w = winding_numbers(verts, gather_tris(verts, faces))
    ext = w <= np.float32(0.99)
    if segments and (always_filter or (~ext).sum() > 0):
        for seg in segments:
            seg_ext = seg.exterior(verts)
            ext[seg.vidx[~seg_ext]] = True
    return ext, w


# ---------------------------------------------------------------- contact terms
def _tanh2_terms(d: np.ndarray, sel: np.ndarray, weight: float, scale: float):
    """sum_sel weight*tanh(d/scale)^2 and d(sum)/dd (zero outside sel)."""
    t = np.tanh(d / np.float32(scale), dtype=np.float32)
    val = (np.float32(weight) * t[sel] ** 2).sum(dtype=np.float64)
    dd = np.zeros_like(d)
    dd[sel] = 2.0 * weight * t[sel] * (1.0 - t[sel] ** 2) / scale
    return float(val), dd


def _pair_distance(verts: np.ndarray, partner: np.ndarray):
    diff = verts - verts[partner]
    d = np.sqrt((diff * diff).sum(1, dtype=np.float32), dtype=np.float32)
    return diff, d


def _scatter_pair_grad(diff, d, dd, partner, num_rows):
    """d = |v_i - v_p(i)|: both endpoints get gradient; norm backward at d == 0 is 0."""
    g = np.zeros((num_rows, 3), np.float64)
    safe = d > 0
    coef = np.zeros_like(d, dtype=np.float64)
    coef[safe] = dd[safe] / d[safe]
    contrib = diff.astype(np.float64) * coef[:, None]
    g += contrib
    np.add.at(g, partner, -contrib)
    return g


def smplify_contact_body(verts, faces, geomask, euclthres, segments=None,
                         region_pairs: Optional[List[Tuple[np.ndarray, np.ndarray]]] = None):
    """One iteration of the per-body loop of contact_fitting_loss, losses.py:74-117.

    Returns dict(contact, r2r, exterior, argmin, d, grad_contact[V,3], grad_r2r[V,3]);
    the caller applies the x10 / contact_loss_weight factors of losses.py:120.
    region_pairs: the (verts1_idxs, verts2_idxs) of every annotated pair with
    gt_contact == 1 (losses.py:110-114), or None when has_discrete_contact is false.
    """
    verts = _c32(verts)
    num_verts = verts.shape[0]
    ext, w = exterior_flags(verts, faces, segments, always_filter=False)
    min_d2, arg = v2v_min_masked(verts, geomask)
    diff, d = _pair_distance(verts, arg)                        # :98
    in_contact = d < np.float32(euclthres)                      # :99
    inside, dd_in = _tanh2_terms(d, ~ext, 1.0, 0.04)            # :100-101
    outside, dd_out = _tanh2_terms(d, ext & in_contact, 0.005, 0.005)  # :102-104
    grad = _scatter_pair_grad(diff, d, dd_in + dd_out, arg, num_verts)
    r2r = 0.0
    grad_r2r = np.zeros((num_verts, 3), np.float64)
    r2r_pairs = []
    if region_pairs is not None:
        gm = np.asarray(geomask, bool)
        for r1, r2 in region_pairs:                             # :110-116
            r1, r2 = np.asarray(r1, np.int64), np.asarray(r2, np.int64)
            sub = pairwise_sq(verts[r1], verts[r2])
            sub = np.where(gm[np.ix_(r1, r2)], sub, np.float32(np.inf))
            k = int(np.argmin(sub))
            a, b = r1[k // len(r2)], r2[k % len(r2)]
            r2r += float(sub.flat[k])
            r2r_pairs.append((int(a), int(b), float(sub.flat[k])))
            dv = 2.0 * (verts[a].astype(np.float64) - verts[b].astype(np.float64))
            grad_r2r[a] += dv
            grad_r2r[b] -= dv
    return dict(contact=inside + outside, inside=inside, outside=outside, r2r=r2r,
                exterior=ext, winding=w, argmin=arg, min_d2=min_d2, d=d,
                grad_contact=grad, grad_r2r=grad_r2r, r2r_pairs=r2r_pairs)


def face_normals(tris: np.ndarray) -> np.ndarray:
    """tuch/train/loss.py:30-41."""
    e0 = tris[:, 1] - tris[:, 0]
    e1 = tris[:, 2] - tris[:, 0]
    n = np.cross(e0, e1).astype(np.float32)
    return n / np.sqrt((n * n).sum(1, keepdims=True, dtype=np.float32), dtype=np.float32)


def train_contact_body(verts, faces, geomask, euclthres, segments, use_hd,
                       hd_idx=None, hd_w=None, hd_face=None, hd_arg_given=None, hd_ext_given=None):
    """One iteration of the per-body loop of RegressorLoss.contact_loss, loss.py:247-315.

    hd_idx/hd_w [N_hd,K]: the non-zeros of each Vert_Regressor row (K = 3 for barycentric samples); hd_face [N_hd]:
    faces_vert_is_sampled_from.  Returns dict(loss, grad[V,3], ...).
    hd_arg_given / hd_ext_given (tests): evaluate the terms with these partners / flags of the selected HD points
    (in the order of np.where(hd_sel)) instead of the ones found here -- to separate "a different pick between
    candidates the reference's float32 cannot tell apart" from an arithmetic difference; ``hd_argmin`` and
    ``hd_exterior`` of the result are always the ones found here.
    """
    verts = _c32(verts)
    faces = _ci64(faces)
    num_verts = verts.shape[0]
    gm = np.asarray(geomask, bool)
    ext, w = exterior_flags(verts, faces, segments, always_filter=True)   # :260-266
    min_d2, arg = v2v_min_masked(verts, gm)                               # :269-270
    out = dict(exterior_verts=ext, winding=w, argmin=arg, min_d2=min_d2)
    if not use_hd:
        diff, d = _pair_distance(verts, arg)                              # :303
        pull, dd_pull = _tanh2_terms(d, ext, 0.005, 0.005)                # :307-308
        push, dd_push = _tanh2_terms(d, ~ext, 1.0, 0.04)                  # :311-312
        grad = _scatter_pair_grad(diff, d, dd_pull + dd_push, arg, num_verts)
        out.update(loss=pull + push, pull=pull, push=push, grad=grad, d=d)
        return out
    # HD path, loss.py:274-301
    cand = (min_d2 < np.float32(euclthres) ** 2) | ~ext                   # :278
    face_sel = cand[faces].any(1)                                         # :279-280
    hd_sel = face_sel[hd_face]                                            # :281
    out.update(hd_sel=hd_sel)
    if hd_sel.sum() == 0:                                                 # :284,:300-301
        out.update(loss=0.0, pull=0.0, push=0.0, grad=np.zeros((num_verts, 3)))
        return out
    idx, wgt = hd_idx[hd_sel], hd_w[hd_sel].astype(np.float32)
    hd = (verts[idx] * wgt[:, :, None]).sum(1, dtype=np.float32)          # :285
    gv = faces[hd_face[hd_sel], 0]                                        # :88 geovec_verts
    hd_mask = gm[np.ix_(gv, gv)]                                          # :289
    _, hd_arg = v2v_min_masked(hd, hd_mask)                               # :288-291
    tris = gather_tris(verts, faces)
    offs = hd + np.float32(0.001) * face_normals(tris)[hd_face[hd_sel]]   # :295-296
    hd_ext = winding_numbers(offs, tris) <= np.float32(0.99)              # :297
    use_arg = hd_arg if hd_arg_given is None else np.asarray(hd_arg_given, np.int64)
    use_ext = hd_ext if hd_ext_given is None else np.asarray(hd_ext_given, bool)
    diff, d = _pair_distance(hd, use_arg)                                 # :299
    pull, dd_pull = _tanh2_terms(d, use_ext, 0.005, 0.005)
    push, dd_push = _tanh2_terms(d, ~use_ext, 1.0, 0.04)
    g_hd = _scatter_pair_grad(diff, d, dd_pull + dd_push, use_arg, hd.shape[0])
    grad = np.zeros((num_verts, 3), np.float64)
    for k in range(idx.shape[1]):                                         # the row's non-zeros (3 for barycentric samples)
        np.add.at(grad, idx[:, k], g_hd * wgt[:, k:k + 1].astype(np.float64))
    out.update(loss=pull + push, pull=pull, push=push, grad=grad, d=d, hd_exterior=hd_ext,
               hd_argmin=hd_arg, hd_points=hd, hd_offset_points=offs, hd_mask=hd_mask)
    return out


def train_contact_loss(verts, valid_fit, faces, geomask, euclthres, segments, use_hd, **hd):
    """RegressorLoss.contact_loss, loss.py:240-317: mean over valid bodies."""
    verts = _c32(verts)
    valid = np.where(np.asarray(valid_fit, bool))[0]
    per_body = np.zeros(verts.shape[0], np.float64)
    grad = np.zeros(verts.shape, np.float64)
    for b in valid:
        r = train_contact_body(verts[b], faces, geomask, euclthres, segments, use_hd, **hd)
        per_body[b] = r['loss']
        grad[b] = r['grad'] / len(valid)
    return float(per_body[valid].mean()), grad, per_body


def contact_from_verts(verts, regions: Dict[str, np.ndarray], pairs) -> np.ndarray:
    """TUCH.contact_from_verts, tuch/train/train_module.py:69-91 -> [B,P] (bmm-form)."""
    verts = _c32(verts)
    out = np.zeros((verts.shape[0], len(pairs)), np.float32)
    for k, (ra, rb) in enumerate(pairs):
        i1, i2 = np.asarray(regions[ra], np.int64), np.asarray(regions[rb], np.int64)
        for b in range(verts.shape[0]):
            out[b, k] = pairwise_sq(verts[b][i1], verts[b][i2]).min()
    return out


def eft_contact_body(verts, faces, geomask, segments, region_pairs):
    """One trip of the per-body loop of EFTLoss.contact_loss, tuch/eft/loss.py:140-179: means instead
    of sums, no distance gate on the exterior term; the caller applies 100 * (contact + 0.5 * r2r)."""
    r = smplify_contact_body(verts, faces, geomask, np.inf, segments, region_pairs)
    verts = _c32(verts)
    ext = exterior_flags(verts, faces, segments, always_filter=True)[0]      # eft/loss.py:149-152 (always)
    diff, d = _pair_distance(verts, r['argmin'])
    n_in, n_out = int((~ext).sum()), int(ext.sum())
    vin, dd_in = _tanh2_terms(d, ~ext, 1.0, 0.04)
    vout, dd_out = _tanh2_terms(d, ext, 0.005, 0.005)
    contact = vin / max(n_in, 1) + vout / max(n_out, 1)
    grad = _scatter_pair_grad(diff, d, dd_in / max(n_in, 1) + dd_out / max(n_out, 1), r['argmin'], verts.shape[0])
    return dict(contact=contact, r2r=r['r2r'], grad_contact=grad, grad_r2r=r['grad_r2r'])
