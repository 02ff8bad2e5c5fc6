"""GPU hardening tests (SURVEY.md §5 sanitiser plan, risk R2): workspace guard words armed around the real workloads,
degenerate geometry against the oracle, a non-finite body inside a batch.

Reference semantics of the degenerate cases: tuch/utils/contact.py:79-109 -- a zero-area triangle contributes
2 atan2(~0, den): 0 where den > 0 (always, for a triangle with two coincident corners: den = 2|B|(|A||B| + A.B) >= 0),
and a query that coincides with a corner of a triangle has A = 0 -> atan2(0, 0) = 0; NaN vertices give a NaN loss.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import golden_io as gio
from helpers import golden, golden_mask, oracle_segments, report, touches_surface
from oracle import contact as oc

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(g, gm, with_segments=True):
    from tuch_amd.ops import ContactModel
    segs = gio.unpack_segments(g)
    seg_list = [(s['vidx'], list(s['bands'].values())) for s in segs.values()] if with_segments else None
    return ContactModel(g['faces'], gm, seg_list, device=torch.device(DEV))


# ------------------------------------------------------------------------------------------------ workspace canaries
def test_canaries_armed_around_the_real_workloads():
    """Option canary = 1 puts 256 guard bytes behind every workspace region of the hot calls (csrc/workspace.h), armed
    before and compared after the call's kernels.  Armed here around: the full-size batch-64 stage-2 step (inside test
    with segments, search, region pairs, tail), the HD training step at batch 64, a batch of 300 (several grid passes)
    and the forced pair-list overflow (block-major fallback).  No guard word may change; the self-test (which overruns
    a region on purpose) must count its hit, so a silent mechanism cannot pass."""
    import bench
    from tuch_amd.smplify.losses import contact_model_for
    dev = torch.device(DEV)
    p = bench.build_problem(64, dev, seed=1002)
    model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
    crit = bench.regressor_loss(p, True)
    models = {id(model): model, id(crit._model): crit._model}
    try:
        for m in models.values():
            m.set_option('canary', 1)
            assert m.get_option('canary') == 1
            assert m.canary_selftest() > 0                  # the mechanism counts an overrun
            assert m.canary_hits() == 0                     # ... and the self-test restored the counter
        step = bench.make_step(p)                           # eager launches: the options apply to every call
        for _ in range(3):
            loss, _ = step()
        assert np.isfinite(float(loss))
        hd_step = bench.make_train_step(p, True)
        for _ in range(2):
            stats = hd_step()
        assert np.isfinite(float(stats[0]))
        plain_step = bench.make_train_step(p, False)
        plain_step()
        torch.cuda.synchronize()
        hits = {k: m.canary_hits() for k, m in models.items()}
        report('canary: guard words changed by the batch-64 stage-2 step + HD / plain training steps', sum(hits.values()), 0)
        assert all(h == 0 for h in hits.values()), hits
    finally:
        for m in models.values():
            m.set_option('canary', 0)


@pytest.mark.parametrize('tag,batch,cap', [('small', 300, 0), ('ico_medium', 130, 0), ('full', 5, 1), ('full2', 5, 2)])
def test_canaries_big_batches_and_pair_list_overflow(tag, batch, cap):
    from test_gpu_contact import _posed_batch
    g, verts = _posed_batch(tag, batch, 23)
    model = _model(g, golden_mask(tag))
    model.set_option('canary', 1)
    if cap:
        model.set_option('ray_pair_cap', cap)
    want = model.canary_selftest()
    assert want > 0
    ext = model.exterior_flags(verts, apply_segments=True)
    ext2, w, _, _ = model.exterior_flags(verts, apply_segments=True, return_details=True)
    mn, part = model.v2v_min(verts)
    rng = np.random.default_rng(3)
    q = min(500, verts.shape[1])
    pts = verts[:, :q] + 0.004 * torch.tensor(rng.standard_normal((batch, q, 3)).astype(np.float32), device=verts.device)
    model.winding_points(verts, pts.contiguous(), flags_only=True)
    model.winding_points(verts, pts.contiguous())
    torch.cuda.synchronize()
    assert model.canary_hits() == 0
    assert int((ext == 0).sum()) > 0 and torch.isfinite(mn).all()


# ------------------------------------------------------------------------------------------------ degenerate geometry
def _flags_vs_oracle(tag, verts_np, what, with_segments=True, allow=0):
    """Flags of the loss path (ray crossings + segment filter) against the oracle's (solid-angle sums as the reference,
    contact.py:79-147, segmentation.py:81-99) on the given bodies.  Returns (mismatches off the threshold, vertices)."""
    g = golden(tag)
    model = _model(g, None, with_segments)
    verts = torch.tensor(verts_np, device=DEV)
    ext = model.exterior_flags(verts, apply_segments=with_segments).cpu().numpy().astype(bool)
    ext_sa = None
    model.set_option('winding_ray', 0)
    ext_sa, w_gpu, _, _ = model.exterior_flags(verts, apply_segments=with_segments, return_details=True)
    ext_sa, w_gpu = ext_sa.cpu().numpy().astype(bool), w_gpu.cpu().numpy()
    osegs = oracle_segments(g) if with_segments else []
    bad = []
    for b in range(verts_np.shape[0]):
        want, w = oc.exterior_flags(verts_np[b], g['faces'], osegs, True)
        clear = np.abs(w - 0.99) > 1e-4
        for seg in osegs:                                   # ... and off the threshold of every segment test it takes part in
            w_s = oc.winding_numbers(verts_np[b][seg.vidx], seg.closed_tris(verts_np[b]))
            clear[seg.vidx[np.abs(w_s - 0.99) <= 1e-4]] = False
        for vid in np.nonzero((ext[b] != want) & clear)[0]:
            bad.append((b, int(vid), float(w[vid]), float(w_gpu[b, vid]), bool(ext_sa[b, vid])))
    report('%s: ray-crossing flags != oracle off the threshold' % what, len(bad), ext.size)
    return bad, ext


def test_collapsed_triangles_coincident_vertices():
    """Bodies with COLLAPSED triangles: one vertex of an edge moved onto the other (two coincident vertices, every face
    on that edge has zero area, the faces around the pair share a corner position).  The reference gives such faces
    2 atan2(~0, den >= 0) = 0 and, for the two coincident vertices as queries, skips the faces around BOTH of them
    (A = 0 -> atan2(0, 0) = 0).  Flags must be the oracle's at every vertex whose winding number is off the threshold;
    the coincident pair itself lies ON the surface of the faces around its twin -- a jump of the winding number, where
    the reference's own float sum is the half-space value (exterior) -- and is checked to be exterior too."""
    tag = 'medium'
    g = golden(tag)
    verts = g['verts'][:2].copy()
    faces = g['faces']
    rng = np.random.default_rng(7)
    edges = faces[rng.choice(len(faces), 12, replace=False)][:, :2]
    pairs = []
    for b in range(verts.shape[0]):
        for u, v in edges[b * 6:(b + 1) * 6]:
            verts[b, v] = verts[b, u]                      # collapse edge (u, v): faces on it become needles of zero area
            pairs.append((b, int(u), int(v)))
    bad, ext = _flags_vs_oracle(tag, verts, 'collapsed triangles [medium]')
    twins = {(b, x) for b, u, v in pairs for x in (u, v)}
    for b, vid, w, w_gpu, e_sa in bad:
        # only a vertex ON another face (coincident with a corner of faces it does not belong to) may differ
        assert (b, vid) in twins or touches_surface(verts[b], faces, vid), (b, vid, w, w_gpu)
    assert len([x for x in bad if (x[0], x[1]) not in twins]) <= 2


def test_zero_area_cap_fan_and_pinched_segment():
    """A segment whose boundary loop is collapsed to ONE point (the limb pinched off): every cap-fan triangle
    [b_{i+1}, b_i, centroid] (segmentation.py:56-66) and every body face with two loop vertices has zero area.  The
    segment filter and the body test must still agree with the oracle off the threshold."""
    tag = 'medium'
    g = golden(tag)
    verts = g['verts'][:2].copy()
    segs = gio.unpack_segments(g)
    name = sorted(segs.keys())[0]
    band = list(segs[name]['bands'].values())[0]
    for b in range(verts.shape[0]):
        verts[b, band] = verts[b, band].mean(0, keepdims=True)          # all loop vertices on the loop's centroid
    bad, ext = _flags_vs_oracle(tag, verts, 'zero-area cap fan [medium, segment %s]' % name)
    loop = set(int(x) for x in band)
    for b, vid, w, w_gpu, e_sa in bad:
        assert vid in loop or touches_surface(verts[b], g['faces'], vid), (b, vid, w, w_gpu)


def test_axis_aligned_duplicate_coordinates():
    """Coordinates snapped to a coarse grid (1/32 m): thousands of vertices share x / y / z values, rays pass through
    edges and vertices all the time (the tie rules of the edge functions), neighbouring vertices coincide and faces
    collapse where the mesh is finer than the grid.  Flags against the oracle off the threshold; a mismatch is allowed
    only where the vertex touches another face (coincident with it: on a jump of the winding number)."""
    tag = 'medium'
    g = golden(tag)
    verts = (np.round(g['verts'][:2] * 32.0) / 32.0).astype(np.float32)
    bad, ext = _flags_vs_oracle(tag, verts, 'grid-snapped body [medium, 1/32 m]', with_segments=False)
    for b, vid, w, w_gpu, e_sa in bad:
        assert touches_surface(verts[b], g['faces'], vid), (b, vid, w, w_gpu)
    # the unsnapped axis-aligned template for scale: exact ties only, no mismatch at all
    verts = (np.round(g['verts'][:1] * 512.0) / 512.0).astype(np.float32)
    bad, _ = _flags_vs_oracle(tag, verts, 'grid-snapped body [medium, 1/512 m]', with_segments=False)
    for b, vid, w, w_gpu, e_sa in bad:
        assert touches_surface(verts[b], g['faces'], vid), (b, vid, w, w_gpu)


# ------------------------------------------------------------------------------------------------ non-finite input
_NAN_CHILD = r'''
import json, sys
import numpy as np
import torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + '/tests')
import bench
from tuch_amd.ops import MODE_SMPLIFY, contact_terms
from tuch_amd.smplify.losses import contact_model_for
dev = torch.device('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
p = bench.build_problem(8, dev, seed=1002)
model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices.contiguous()

def run(v):
    v = v.clone().requires_grad_(True)
    ext, mn, partner, _ = model.exterior_and_partner(v.detach(), apply_segments=True)
    per_body, _ = contact_terms(v, partner, ext, None, MODE_SMPLIFY, 0.02)
    per_body.sum().backward()
    torch.cuda.synchronize()
    return ext.cpu().numpy(), mn.cpu().numpy(), partner.cpu().numpy(), per_body.detach().cpu().numpy(), v.grad.cpu().numpy()

from tuch_amd import ops
assert ops.deterministic()           # the default: gradient scatters through integer atomics, bit-reproducible, so "identical" can be asked
clean = run(verts)
again = run(verts)
out = {'repeat': {'others_bit_identical': all(np.array_equal(a, c) for a, c in zip(again, clean)), 'differs': []}}
names = ('exterior', 'min_d2', 'partner', 'contact', 'grad')
for kind, value, ids in (('nan_one_vertex', float('nan'), [1234]), ('inf_one_vertex', float('inf'), [77]),
                         ('nan_whole_body', float('nan'), None), ('nan_scattered', float('nan'), list(range(0, 6890, 13)))):
    v = verts.clone()
    if ids is None:
        v[3] = value
    else:
        v[3, ids, 1] = value
    got = run(v)
    others = [b for b in range(8) if b != 3]
    differs = [n for n, a, c in zip(names, got, clean) if not np.array_equal(a[others], c[others])]
    V = verts.shape[1]
    out[kind] = {'others_bit_identical': not differs, 'differs': differs, 'bad_body_loss': float(got[3][3]),
                 'bad_body_loss_finite': bool(np.isfinite(got[3][3])),
                 'partners_in_range': bool(((got[2] >= 0) & (got[2] < V)).all())}
# the whole stage-2 step (LBS + objective + backward + Adam) with a NaN pose: must terminate
p['body_pose'][5, 7] = float('nan')
step = bench.make_step(p)
for _ in range(2):
    loss, _ = step()
torch.cuda.synchronize()
out['step_with_nan_pose'] = {'loss_finite': bool(np.isfinite(float(loss)))}
print('RESULT ' + json.dumps(out), flush=True)
'''


def test_nonfinite_body_in_a_batch_of_eight():
    """One body of a batch of 8 carries NaN / Inf coordinates (one vertex, scattered vertices, the whole body).  Runs
    in a child process under a timeout (a hang must fail the test, not the session): the call returns; the other seven
    bodies' flags, minima, partners, contact values and gradients are BIT-identical to a run without the bad body; every
    partner index stays in range; the bad body's contact value is non-finite, as the reference's would be
    (tuch/utils/contact.py:79-109: NaN propagates through the distance matrix and the solid angles into the loss)."""
    import json
    env = dict(os.environ)
    res = subprocess.run([sys.executable, '-c', _NAN_CHILD % {'root': ROOT}], env=env, capture_output=True, text=True,
                         timeout=420)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    line = [l for l in res.stdout.splitlines() if l.startswith('RESULT ')][-1]
    out = json.loads(line[len('RESULT '):])
    for kind, r in out.items():
        if kind == 'step_with_nan_pose':
            assert not r['loss_finite'], out
            continue
        assert r['others_bit_identical'], (kind, r)
        if kind == 'repeat':
            continue
        assert r['partners_in_range'], (kind, out)
        # NaN coordinates: the reference's d_i = |v_i - v_j*| of the bad vertex is NaN -> NaN loss (losses.py:98-105).  An INF
        # coordinate gives the reference a finite (meaningless) value as well -- tanh(inf)^2 = 1 -- so only NaN is held to NaN
        if kind.startswith('nan'):
            assert not r['bad_body_loss_finite'], (kind, out)
        report('non-finite input [%s]: other bodies bit-identical, bad body loss %r' % (kind, r['bad_body_loss']), 0, 7)
