"""Shared test helpers: golden loading and oracle-side objects built from a fixture."""
from __future__ import annotations

import functools

import numpy as np

import golden_io as gio
from oracle import contact as oc


@functools.lru_cache(maxsize=None)
def golden(tag: str):
    data = gio.load('contact_%s.npz' % tag)
    return {k: data[k] for k in data.files}


@functools.lru_cache(maxsize=None)
def golden_mask(tag: str) -> np.ndarray:
    return gio.unpack_mask(golden(tag))


def oracle_segments(g):
    segs = gio.unpack_segments(g)
    return [oc.Segment(n, g['faces'], s['vidx'], list(s['bands'].values())) for n, s in segs.items()]


def region_pair_lists(g, body_index: int):
    """(verts1_idxs, verts2_idxs) of the pairs annotated for body b (losses.py:110-114)."""
    regions, pairs = gio.unpack_regions(g)
    gt = g['gt_contact'][body_index]
    return [(regions[a], regions[b]) for k, (a, b) in enumerate(pairs) if gt[k] == 1]


def assert_close(actual, expected, rtol, atol, what=''):
    actual = np.asarray(actual, np.float64)
    expected = np.asarray(expected, np.float64)
    err = np.abs(actual - expected)
    tol = atol + rtol * np.abs(expected)
    if not np.all(err <= tol):
        k = int(np.argmax(err - tol))
        raise AssertionError('%s: max violation at flat index %d: got %r want %r (|err|=%g tol=%g)'
                             % (what, k, actual.flat[k], expected.flat[k], err.flat[k], tol.flat[k]))


def close_logged(actual, expected, rtol, atol, what=''):
    """assert_close that also logs what was OBSERVED (gpurun_out/parity_counts.txt): the largest error relative to the
    largest reference entry and relative to the bound -- the loop-level tolerances are set from these."""
    a, e = np.asarray(actual, np.float64), np.asarray(expected, np.float64)
    err = np.abs(a - e)
    bound = atol + rtol * np.abs(e)
    _log('loop %-64s max|err|/max|ref| %.2e   max err/bound %.3f' % (what, err.max() / max(np.abs(e).max(), 1e-30),
                                                                     (err / np.maximum(bound, 1e-300)).max()))
    assert_close(actual, expected, rtol, atol, what)


def touches_surface(verts_b, faces, vid, tol=2e-6):
    """float64: does vertex `vid` lie within `tol` of a triangle it is not a corner of (inside its outline)?
    There the winding number jumps by one across the triangle: the reference's float32 sum, any other summation
    order and a crossing count may legitimately land on either side."""
    v = np.asarray(verts_b, np.float64)
    faces = np.asarray(faces)
    tri = v[faces[~(faces == vid).any(1)]]
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    n = np.cross(b - a, c - a)
    nn = np.linalg.norm(n, axis=1)
    ok = nn > 0
    n = n[ok] / nn[ok, None]
    a, b, c = a[ok], b[ok], c[ok]
    p = v[vid]
    dist = ((p - a) * n).sum(1)
    near = np.abs(dist) < tol
    if not near.any():
        return False
    a, b, c, n, dist = a[near], b[near], c[near], n[near], dist[near]
    q = p - dist[:, None] * n
    inside = np.ones(len(a), bool)
    for s, e in ((a, b), (b, c), (c, a)):
        inside &= (np.cross(e - s, q - s) * n).sum(1) >= -tol * np.linalg.norm(e - s, axis=1)
    return bool(inside.any())


def _log(line: str) -> None:
    import os
    print(line)
    try:
        out = os.path.join(gio.GOLDEN_DIR, '..', '..', 'gpurun_out')
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_counts.txt'), 'a') as f:
            f.write(line + '\n')
    except OSError:
        pass


def report_value(what, value):
    """An observed maximum (printed with -s, appended to gpurun_out/parity_counts.txt): what the tolerances are set from."""
    _log('%-72s %.3e' % (what, value))


def report(what, count, total):
    """Observed index/flag mismatches against the reference (allowed only between candidates tied within the
    reference's own float32 noise): printed (-s) and appended to gpurun_out/parity_counts.txt."""
    _log('%-72s %6d / %d' % (what, count, total))


# Relative tolerance of gradients against the reference's autograd (north_star: 1e-4), on top of an absolute floor that
# is a fraction of the largest entry (cancellation in float32 sums).  Observed maxima are logged by grad_close.
GRAD_RTOL = 1e-4
# see tests/test_oracle_golden.py: the quantised derivative of a saturated tanh^2 term in float32 autograd
TANH_QUANTUM = 12 * 2.4e-7


def grad_close(actual, expected, floor, what, rtol=GRAD_RTOL, quantum=False):
    """|actual - expected| <= rtol |expected| + floor * max|expected| (+ TANH_QUANTUM for the HD term); logs the
    observed maximum of the error relative to the largest entry and relative to that bound."""
    actual, expected = np.asarray(actual, np.float64), np.asarray(expected, np.float64)
    scale = np.abs(expected).max()
    atol = floor * scale + (TANH_QUANTUM if quantum else 0.0)
    err = np.abs(actual - expected)
    _log('grad %-60s max|err|/max|ref| %.2e   max err/bound %.3f' % (what, err.max() / max(scale, 1e-30),
                                                                     (err / (atol + rtol * np.abs(expected))).max()))
    assert_close(actual, expected, rtol, atol, what)


def point_touches_surface(p, verts_b, faces, tol=2e-6):
    """float64: does the point lie within `tol` of a triangle of the mesh, inside its outline?  There the winding number
    of the point jumps by one: which side a float32 evaluation lands on is arbitrary (for the reference's sum too)."""
    v = np.asarray(verts_b, np.float64)
    tri = v[np.asarray(faces)]
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    n = np.cross(b - a, c - a)
    nn = np.linalg.norm(n, axis=1)
    ok = nn > 0
    n, a, b, c = n[ok] / nn[ok, None], a[ok], b[ok], c[ok]
    p = np.asarray(p, np.float64)
    dist = ((p - a) * n).sum(1)
    near = np.abs(dist) < tol
    if not near.any():
        return False
    a, b, c, n, dist = a[near], b[near], c[near], n[near], dist[near]
    q = p - dist[:, None] * n
    inside = np.ones(len(a), bool)
    for s0, e0 in ((a, b), (b, c), (c, a)):
        inside &= (np.cross(e0 - s0, q - s0) * n).sum(1) >= -tol * np.linalg.norm(e0 - s0, axis=1)
    return bool(inside.any())


def hd_picks_vs_oracle(hd_model, saved, batch, b, verts_b, faces, geomask, euclthres, osegs, hd_idx, hd_w, hd_face, what):
    """The HD branch's per-point decisions for body b against the oracle's (loss.py:274-301): same selected set; a
    different partner only between candidates whose squared distances tie within the reference's bmm-form noise (2e-6);
    a different inside/outside flag only for an offset point that touches a triangle.  Returns (oracle result with its
    own picks, oracle result evaluated with the device's picks) -- identical objects when no pick differs."""
    r = oc.train_contact_body(verts_b, faces, geomask, euclthres, osegs, True, hd_idx=hd_idx, hd_w=hd_w, hd_face=hd_face)
    counts, sel = hd_model.selection(saved, batch)
    part, ext = hd_model.details(saved, batch)
    n = int(counts[b])
    want = np.where(r['hd_sel'])[0]
    ids = sel[b, :n].astype(np.int64)
    assert np.array_equal(np.sort(ids), want), (what, n, len(want))
    if n == 0:
        return r, r
    slot_of = np.argsort(ids)                       # oracle position i (caller id want[i]) lives in slot slot_of[i]
    gpu_arg = np.searchsorted(want, part[b, :n][slot_of].astype(np.int64))
    gpu_ext = ext[b, :n][slot_of]
    arg_diff = np.where(gpu_arg != r['hd_argmin'])[0]
    ext_diff = np.where(gpu_ext != r['hd_exterior'])[0]
    report('%s: HD partners != oracle (ties within bmm noise)' % what, len(arg_diff), n)
    report('%s: HD inside/outside flags != oracle (points touching a triangle)' % what, len(ext_diff), n)
    hd = r['hd_points'].astype(np.float64)
    for i in arg_diff:
        d_gpu = ((hd[i] - hd[gpu_arg[i]]) ** 2).sum()
        d_ref = ((hd[i] - hd[r['hd_argmin'][i]]) ** 2).sum()
        assert abs(d_gpu - d_ref) < 2e-6 and r['hd_mask'][gpu_arg[i], i], (what, i, d_gpu, d_ref)
    for i in ext_diff:
        assert point_touches_surface(r['hd_offset_points'][i], verts_b, faces), (what, i)
    assert len(arg_diff) <= max(3, n // 500) and len(ext_diff) <= 3
    if len(arg_diff) == 0 and len(ext_diff) == 0:
        return r, r
    r2 = oc.train_contact_body(verts_b, faces, geomask, euclthres, osegs, True, hd_idx=hd_idx, hd_w=hd_w, hd_face=hd_face,
                               hd_arg_given=gpu_arg, hd_ext_given=gpu_ext)
    return r, r2
