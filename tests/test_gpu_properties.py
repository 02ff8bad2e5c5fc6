"""GPU: size-independent properties at BASELINE.json's full size (batch 64, V=6890, F=13776) and
edge cases (ragged sizes, empty tables, fully masked columns, ignored bodies)."""
import numpy as np
import pytest
import torch

from helpers import assert_close
from oracle import contact as oc
from oracle import lbs as ol
from synthetic import make_body, random_poses

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def full():
    body = make_body(84, 82, with_geodesics=False)
    bp, go, be = random_poses(64, 4242)
    v, _ = ol.smpl_forward(ol.model_tensors(body), torch.tensor(be), torch.tensor(bp), torch.tensor(go))
    idx = np.arange(body.num_verts)
    mask = np.abs(idx[:, None] - idx[None, :]) > 300          # any symmetric mask will do here
    return body, v.numpy().astype(np.float32), mask


def test_full_batch_winding_is_deterministic_and_rigid_invariant(full):
    from tuch_amd.ops import ContactModel
    body, verts_np, _ = full
    model = ContactModel(body.faces, None, None, device=DEV)
    verts = torch.tensor(verts_np, device=DEV)
    e1, w1, _, _ = model.exterior_flags(verts, apply_segments=False, return_details=True)
    e2, w2, _, _ = model.exterior_flags(verts, apply_segments=False, return_details=True)
    assert torch.equal(w1, w2) and torch.equal(e1, e2)                      # bit-reproducible
    # rigid motion leaves winding numbers unchanged (up to rounding of the moved coordinates)
    rot = ol.rodrigues(torch.tensor([[0.4, -0.7, 0.2]])).to(DEV)[0]
    moved = verts @ rot.T + torch.tensor([0.3, -0.2, 0.5], device=DEV)
    _, w3, _, _ = model.exterior_flags(moved.contiguous(), apply_segments=False, return_details=True)
    err = (w3 - w1).abs()
    assert err.median() < 1e-6 and torch.quantile(err.flatten()[::7], 0.999) < 5e-5
    # spot check against the CPU oracle
    for b in (0, 37, 63):
        wo = oc.winding_numbers(verts_np[b], oc.gather_tris(verts_np[b], body.faces))
        d = np.abs(w1[b].cpu().numpy() - wo)
        assert np.percentile(d, 99) < 5e-6 and d.max() < 2e-4


def test_full_batch_v2v_matches_oracle_and_is_deterministic(full):
    from tuch_amd.ops import ContactModel
    body, verts_np, mask = full
    model = ContactModel(body.faces, mask, None, device=DEV)
    verts = torch.tensor(verts_np, device=DEV)
    mn1, a1 = model.v2v_min(verts)
    mn2, a2 = model.v2v_min(verts)
    assert torch.equal(mn1, mn2) and torch.equal(a1, a2)
    a1n, mn1n = a1.cpu().numpy().astype(np.int64), mn1.cpu().numpy()
    assert mask[a1n, np.arange(mask.shape[0])[None, :]].all()              # partners respect the mask
    for b in (0, 31, 63):
        v = verts_np[b].astype(np.float64)
        d_true = ((v - v[a1n[b]]) ** 2).sum(1)
        assert_close(mn1n[b], d_true, 1e-5, 1e-9, 'min = distance to the returned partner')
        mo, ao = oc.v2v_min_masked(verts_np[b], mask)
        same = ao == a1n[b]
        assert same.mean() > 0.99
        d_ref = ((v - v[ao]) ** 2).sum(1)
        assert np.all(d_true <= d_ref + 1e-9)                              # never worse than the bmm-form pick


def test_closed_mesh_inside_outside_points(full):
    from tuch_amd import ops
    body, _, _ = full
    v = torch.tensor(body.v_template, device=DEV)[None]
    tris = ops.gather_triangles(v, torch.tensor(body.faces.astype(np.int32), device=DEV))
    pts = torch.tensor([[[0.0, 0.0, 0.0], [0.0, 0.3, 0.0], [3.0, 3.0, 3.0], [0.0, -2.0, 0.5]]], device=DEV)
    w = ops.winding_numbers(pts, tris)[0].cpu().numpy()
    assert_close(w, [1.0, 1.0, 0.0, 0.0], 0, 2e-5, 'inside/outside')


@pytest.mark.parametrize('rings,segs,batch', [(5, 13, 1), (9, 7, 3), (12, 14, 5)])
def test_ragged_sizes_match_oracle(rings, segs, batch):
    """V = rings*segs+2 is not a multiple of 64 (or of anything convenient)."""
    from tuch_amd.ops import ContactModel, contact_terms, MODE_TRAIN
    body = make_body(rings, segs, relax_iters=20)
    bp, go, be = random_poses(batch, 77)
    verts_np = ol.smpl_forward(ol.model_tensors(body), torch.tensor(be), torch.tensor(bp),
                               torch.tensor(go))[0].numpy().astype(np.float32)
    gm = body.geodesics > 0.3
    segs_t = [(s['vidx'], list(s['bands'].values())) for s in body.segments.values()]
    model = ContactModel(body.faces, gm, segs_t or None, device=DEV)
    verts = torch.tensor(verts_np, device=DEV, requires_grad=True)
    ext = model.exterior_flags(verts, apply_segments=bool(segs_t))
    mn, arg = model.v2v_min(verts)
    per_body, _ = contact_terms(verts, arg, ext, None, MODE_TRAIN, 0.02)
    per_body.sum().backward()
    osegs = [oc.Segment(n, body.faces, s['vidx'], list(s['bands'].values())) for n, s in body.segments.items()]
    for b in range(batch):
        r = oc.train_contact_body(verts_np[b], body.faces, gm, 0.02, osegs, False)
        flips = (ext[b].cpu().numpy().astype(bool) != r['exterior_verts'])
        near = np.abs(r['winding'] - 0.99) < 1e-4
        assert not (flips & ~near).any()
        if not flips.any():
            assert_close(per_body[b].item(), r['loss'], 1e-4, 1e-6, 'loss')
            assert_close(verts.grad[b].cpu().numpy(), r['grad'], 1e-3, 1e-5 * max(np.abs(r['grad']).max(), 1e-3), 'grad')


def test_fully_masked_columns_and_ignored_bodies():
    from tuch_amd.ops import ContactModel, contact_terms, MODE_SMPLIFY
    body = make_body(10, 12)
    gm = body.geodesics > 0.3
    gm[:, 5] = False
    gm[5, :] = False                                   # vertex 5 has no admissible partner
    verts = torch.tensor(body.v_template, device=DEV)[None].repeat(2, 1, 1).contiguous()
    model = ContactModel(body.faces, gm, None, device=DEV)
    mn, arg = model.v2v_min(verts)
    assert torch.isinf(mn[:, 5]).all() and (arg[:, 5] == 0).all()          # torch.min/argmin of an all-inf column
    ext = model.exterior_flags(verts, apply_segments=False)
    assert ext.all()                                                        # rest pose: nothing is inside
    valid = torch.tensor([1, 0], dtype=torch.uint8, device=DEV)
    per_body, _ = contact_terms(verts, arg, ext, valid, MODE_SMPLIFY, 0.02)
    assert per_body[1].item() == 0.0                                        # ignored body contributes nothing


def test_model_without_optional_tables():
    from tuch_amd import _C
    from tuch_amd.ops import ContactModel
    body = make_body(10, 12, with_geodesics=False)
    model = ContactModel(body.faces, None, None, None, None, device=DEV)
    verts = torch.tensor(body.v_template, device=DEV)[None]
    assert model.exterior_flags(verts, apply_segments=True).shape == (1, body.num_verts)
    with pytest.raises(_C.TuchError):
        model.v2v_min(verts)
    with pytest.raises(_C.TuchError):
        model.region_pair_min(verts)
    with pytest.raises(_C.TuchError):
        model.exterior_flags(verts.cpu())


# ---- the shape bench.py times: batch 64, V=6890, bench.build_problem's own vertices ---------------------------------
@pytest.fixture(scope='module')
def headline():
    import bench
    dev = torch.device(DEV)
    p = bench.build_problem(64, dev, seed=1002)            # rank 0's problem of the default bench run
    with torch.no_grad():
        verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices.contiguous()
    return p, verts


def _oracle_bodies(p, verts_np, which):
    body = p['body']
    gm = body.geodesics > 0.3
    osegs = [oc.Segment(n, body.faces, s['vidx'], list(s['bands'].values())) for n, s in body.segments.items()]
    return {b: oc.smplify_contact_body(verts_np[b], body.faces, gm, 0.02, osegs, None) for b in which}, gm, osegs


def test_headline_batch64_flags_partners_and_contact_value_for_every_body(headline):
    """What the timed step computes behind the body model -- exterior flags by ray crossings WITH the segment filter,
    masked partners, the contact value of losses.py:96-105 -- at the bench's own launch shape (64 bodies: eight per XCD
    column, choose_v2v_frontier(64), one grid pass of the crossing kernel) against the oracle for ALL 64 bodies."""
    from helpers import report
    from tuch_amd.ops import MODE_SMPLIFY, contact_terms
    from tuch_amd.smplify.losses import contact_model_for
    p, verts = headline
    model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
    assert model.get_option('winding_ray') == 1
    v = verts.clone().requires_grad_(True)
    ext, mn, partner, _ = model.exterior_and_partner(v.detach(), apply_segments=True)     # as _Stage2Tail calls it
    per_body, _ = contact_terms(v, partner, ext, None, MODE_SMPLIFY, 0.02)
    per_body.sum().backward()
    torch.cuda.synchronize()
    verts_np = verts.cpu().numpy()
    ref, gm, _ = _oracle_bodies(p, verts_np, range(64))
    ext_np, part_np, mn_np = ext.cpu().numpy().astype(bool), partner.cpu().numpy().astype(np.int64), mn.cpu().numpy()
    flag_mismatch = part_mismatch = interior = 0
    worst_val = worst_grad = 0.0
    for b in range(64):
        r = ref[b]
        # the SMPLify form applies the segment filter only when the body has an interior vertex (losses.py:85); the
        # device applies it always -- identical, since the filter can only turn interior vertices exterior
        clear = np.abs(r['winding'] - 0.99) > 1e-4
        bad = (ext_np[b] != r['exterior']) & clear
        flag_mismatch += int(bad.sum())
        interior += int((~r['exterior']).sum())
        assert_close(mn_np[b], r['min_d2'], 0, 1e-6, 'min d2 body %d' % b)
        diff = part_np[b] != r['argmin']
        part_mismatch += int(diff.sum())
        vb = verts_np[b].astype(np.float64)
        d_ours = ((vb - vb[part_np[b]]) ** 2).sum(1)
        d_ref = ((vb - vb[r['argmin']]) ** 2).sum(1)
        assert np.all(np.abs(d_ours - d_ref)[diff] < 2e-6)             # different partner only between tied rows
        assert gm[part_np[b], np.arange(gm.shape[0])].all()
        if not (ext_np[b] != r['exterior']).any() and not diff.any():
            got = per_body[b].item()
            worst_val = max(worst_val, abs(got - r['contact']) / max(abs(r['contact']), 1e-6))
            assert_close(got, r['contact'], 1e-4, 1e-6, 'contact value body %d' % b)
            gerr = np.abs(v.grad[b].cpu().numpy() - r['grad_contact'])
            scale = np.abs(r['grad_contact']).max()
            worst_grad = max(worst_grad, gerr.max() / max(scale, 1e-12))
            assert_close(v.grad[b].cpu().numpy(), r['grad_contact'], 1e-4, 2e-6 * scale, 'contact grad body %d' % b)
    report('headline B=64: exterior flags (ray + segments) != oracle where |w-0.99|>1e-4', flag_mismatch, 64 * verts.shape[1])
    report('headline B=64: partners != oracle (tied rows only)', part_mismatch, 64 * verts.shape[1])
    report('headline B=64: interior vertices in the batch', interior, 64 * verts.shape[1])
    report('headline B=64: worst contact value rel err x1e9', int(worst_val * 1e9), 64)
    report('headline B=64: worst contact grad err / max|grad| x1e9', int(worst_grad * 1e9), 64)
    assert flag_mismatch == 0 and part_mismatch <= 64 and interior > 2000


def test_headline_batch64_hd_contact_loss_against_the_oracle(headline):
    """RegressorLoss.contact_loss(use_hd=True) on the bench's vertices at batch 64; eight of the bodies (every eighth:
    one per XCD column) are checked against the oracle's HD branch (loss.py:274-315), all N_hd = 41 328 points."""
    import bench
    from helpers import grad_close, hd_picks_vs_oracle
    p, verts = headline
    body = p['body']
    crit = bench.regressor_loss(p, True)
    which = list(range(0, 64, 8))
    v = verts.clone().requires_grad_(True)
    valid = torch.zeros(64, dtype=torch.bool, device=v.device)
    valid[which] = True
    loss = crit.contact_loss(v, valid)          # mean over the 8 valid bodies, the other 56 ride along in the batch
    loss.backward()
    verts_np = verts.cpu().numpy()
    gm = body.geodesics > 0.3
    osegs = [oc.Segment(n, body.faces, s['vidx'], list(s['bands'].values())) for n, s in body.segments.items()]
    want, grads = [], {}
    saved = crit._hd.last_saved
    for b in which:
        # r: the oracle with its own picks; r2: the oracle's formulas with the device's picks (differences are checked
        # to be ties / touching points inside hd_picks_vs_oracle)
        r, r2 = hd_picks_vs_oracle(crit._hd, saved, 64, b, verts_np[b], body.faces, gm, 0.02, osegs, body.hd_bary_idx,
                                   body.hd_bary_w, body.hd_face_id, 'headline HD body %d' % b)
        want.append(r2['loss'])
        grads[b] = r2['grad'] / len(which)
        assert abs(r2['loss'] - r['loss']) <= 2e-3 * abs(r['loss'])
    assert_close(loss.item(), float(np.mean(want)), 1e-4, 0, 'HD contact loss, 8 of 64 bodies')
    g = v.grad.cpu().numpy()
    assert np.all(g[[b for b in range(64) if b not in which]] == 0)
    for b in which:
        grad_close(g[b], grads[b], 5e-6, 'headline HD grad body %d' % b, quantum=True)


def test_headline_batch64_hd_search_on_the_matrix_cores_against_the_exact_kernel(headline):
    """The HD branch at the bench's batch 64 with ALL bodies valid, once with the matrix-core search (hd_search.hip, the
    default) and once with the exact kernel (v2v_indexed_kernel): same selection, partners identical except between rows
    that tie (verified per point: squared distances within 2e-6 relative + 2e-8), same inside / outside flags, same
    loss to 1e-6 -- for every one of the ~390 000 selected points of the batch."""
    import bench
    from helpers import report
    p, verts = headline
    crit = bench.regressor_loss(p, True)
    model = crit._model
    valid = torch.ones(64, dtype=torch.bool, device=verts.device)
    res = {}
    for form in (1, 0):
        model.set_option('hd_search', form)
        with torch.no_grad():
            loss = crit.contact_loss(verts, valid)
        torch.cuda.synchronize()
        counts, sel = crit._hd.selection(crit._hd.last_saved, 64)
        part, ext = crit._hd.details(crit._hd.last_saved, 64)
        res[form] = (float(loss), counts.copy(), sel.copy(), part.copy(), ext.copy())
    model.set_option('hd_search', 1)
    (l1, c1, s1, p1, e1), (l0, c0, s0, p0, e0) = res[1], res[0]
    assert np.array_equal(c1, c0) and np.array_equal(s1, s0) and np.array_equal(e1, e0)
    assert abs(l1 - l0) <= 1e-6 * abs(l0)
    body = p['body']
    idx, w = np.asarray(body.hd_bary_idx), np.asarray(body.hd_bary_w, np.float64)
    verts_np = verts.cpu().numpy().astype(np.float64)
    differ = total = 0
    for b in range(64):
        n = int(c1[b])
        total += n
        d = np.where(p1[b, :n] != p0[b, :n])[0]
        differ += len(d)
        if len(d) == 0:
            continue
        pt = lambda ids: (verts_np[b][idx[ids]] * w[ids][..., None]).sum(1)      # HD points by caller index
        me, a, c = pt(s1[b, :n][d]), pt(p1[b, :n][d]), pt(p0[b, :n][d])
        da, dc = ((me - a) ** 2).sum(1), ((me - c) ** 2).sum(1)
        assert (np.abs(da - dc) <= 2e-6 * dc + 2e-8).all(), (b, float(np.abs(da - dc).max()))
    report('headline HD: partners, matrix-core search != exact kernel (ties within the key)', differ, total)
    assert differ <= max(8, total // 5000)


def test_headline_batch64_step_reproduces_bit_for_bit_by_default(headline):
    """SURVEY 8(b) asks for a deterministic path: since round 6 it is the default (TUCH_DETERMINISTIC=0 opts out).  The
    bench's own step -- SMPL forward, stage-2 objective, backward, Adam -- captured and replayed eight times from the same
    start, twice over: identical bits in the parameters and in the reported loss."""
    import bench
    from tuch_amd import ops
    p, _ = headline
    assert ops.deterministic()

    def run():
        with ops.off_default_stream(DEV):
            step = bench.capture(bench.make_step(p), 3)
            for _ in range(8):
                loss, _ = step()
            torch.cuda.synchronize()
            value, verts, joints, pose = step.objective()
        return loss.clone(), pose, verts.clone()
    a, b = run(), run()
    for x, y, what in zip(a, b, ('loss of the last step', 'body pose after the steps', 'posed vertices')):
        assert torch.equal(x, y), what
    assert float((a[1] - p['body_pose']).abs().max()) > 1e-3          # the steps did move the parameters


def test_headline_batch64_capped_search_gives_the_same_fit_and_the_same_partners_where_they_count(headline):
    """Round 6: in SMPLify-DC's stage 2 the nearest-vertex search of the vertices the previous iteration found OUTSIDE the body
    starts at the loss's cap (euclthres: an exterior vertex contributes only within it, tuch/smplify/losses.py:99-104) and the
    vertices this iteration's inside test finds inside after all are searched again exhaustively (ops.ContactModel.
    exterior_and_partner(cap=...), option v2v_cap).  (1) Ten iterations of the bench's own step with and without it: the
    same loss and the same parameters, BIT FOR BIT (deterministic gradient sums).  (2) On moving bodies, call after call:
    flags identical, and for every vertex that is inside or has a partner within the cap the capped call returns the plain
    call's (min_d2, partner); for the others min_d2 = cap^2 and the partner is an admissible vertex farther than the cap."""
    import bench
    from helpers import report
    from tuch_amd import ops
    from tuch_amd.smplify.losses import contact_model_for
    p, verts = headline
    model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
    model.set_option('v2v_cap', 1)                   # (off by default: measured, exact, not faster -- csrc/model.h)
    assert model.v2v_can_cap(64)

    def fit(cap):
        model.set_option('v2v_cap', cap)
        with ops.off_default_stream(DEV):
            step = bench.make_step(p)
            losses = []
            for _ in range(10):
                losses.append(step()[0].clone())
            torch.cuda.synchronize()
            return torch.stack(losses), step.objective()[3]
    try:
        la, pa = fit(1)
        lb, pb = fit(0)
    finally:
        model.set_option('v2v_cap', 1)
    assert torch.equal(la, lb) and torch.equal(pa, pb)
    try:
        _capped_outputs(model, p, verts, report)
    finally:
        model.set_option('v2v_cap', 0)


def _capped_outputs(model, p, verts, report):
    from tuch_amd import ops
    # (2) the raw outputs on bodies that move between calls (what the hints and the predicted flags are for)
    cap = 1.001 * 0.02 + 1e-6
    gm = p['body'].geodesics > 0.3
    rng = np.random.default_rng(3)
    drift = torch.tensor(rng.standard_normal((64, 1, 3)).astype(np.float32) * 0.0, device=verts.device)
    wobble = torch.tensor(rng.standard_normal(tuple(verts.shape)).astype(np.float32), device=verts.device)
    capped_cols = fixed_cols = 0
    for it in range(4):
        v = (verts + 0.002 * it * wobble + drift).contiguous()
        with ops.off_default_stream(DEV):
            ext_c, mn_c, arg_c, _ = model.exterior_and_partner(v, apply_segments=True, iterative=True, cap=cap)
            ext_p, mn_p, arg_p, _ = model.exterior_and_partner(v, apply_segments=True, iterative=True)
        torch.cuda.synchronize()
        assert torch.equal(ext_c, ext_p)
        body_inside = model.exterior_flags(v, apply_segments=False) == 0
        # what the loss looks at: the vertices that are inside (final flags) or have a partner within the cap
        counts = ((ext_c == 0) | (mn_p < cap * cap))
        assert torch.equal(mn_c[counts], mn_p[counts]) and torch.equal(arg_c[counts], arg_p[counts])
        rest = ~counts
        # the others: either searched without a cap (predicted inside; every vertex in the first call) -> the plain result,
        cut = rest & (mn_c == np.float32(cap) * np.float32(cap))
        plain = rest & ~cut
        assert torch.equal(mn_c[plain], mn_p[plain]) and torch.equal(arg_c[plain], arg_p[plain])
        # or cut off at the cap: the partner is a real admissible vertex farther than the cap
        d2 = ((v - torch.gather(v, 1, arg_c.long()[..., None].expand(-1, -1, 3))) ** 2).sum(-1)
        assert bool((d2[cut] >= np.float32(cap) ** 2 * (1 - 1e-5)).all())
        gm_t = torch.tensor(gm, device=v.device)
        idx = torch.arange(v.shape[1], device=v.device)[None].expand(64, -1)
        assert bool(gm_t[idx[cut], arg_c.long()[cut]].all())
        capped_cols += int(cut.sum())
        fixed_cols += int((body_inside & ~(mn_p < cap * cap)).sum())
    report('capped search: vertices cut off at the cap (4 calls x 64 bodies)', capped_cols, 4 * 64 * verts.shape[1])
    report('capped search: inside vertices beyond the cap (searched without one / again)', fixed_cols, 4 * 64 * verts.shape[1])
    assert capped_cols > 0.5 * 3 * 64 * verts.shape[1]
