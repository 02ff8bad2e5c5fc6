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
