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
