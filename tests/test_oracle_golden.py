"""Pin the CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only.

Tolerances: quantities that go through the reference's bmm-form squared distance
(contact.py:27-42) carry ~1e-7*|x|^2 of BLAS-order noise -> absolute 1e-6; everything
else is float32 round-off of identical formulas -> 1e-5 absolute / 1e-5 relative.
"""
import numpy as np
import pytest

import golden_io as gio
from helpers import assert_close, golden, golden_mask, oracle_segments, region_pair_lists
from oracle import contact as oc

TAGS = ['small', 'medium', 'ico_small', 'ico_medium']   # ico_*: irregular topology (valence 4-9, ragged painted segments)
# float32 autograd evaluates d tanh^2 as 2 t (1 - t*t): for a saturated term (exterior point a few cm from its partner,
# d / 0.005 > 6) 1 - t*t is a multiple of 2^-23, so one ulp of difference in tanh moves a point's derivative by 2.4e-7
# whatever its size; a vertex collects ~12 such HD-point terms (weights 1/3 ... 2/3, own and partner side)
TANH_QUANTUM = 12 * 2.4e-7
FULL = ['full', 'full2', 'ico_full']                      # SMPL-sized; full2: forearm through the trunk + an ignored body


def test_pairwise_dense():
    g = golden('small')
    p = oc.pairwise_sq(g['verts'][0], g['verts'][0])
    assert_close(p, g['pairwise_b0'], 0, 1e-6, 'pairwise')
    assert np.all(np.diag(p) == 0)


def test_pairwise_adjoint_vs_reference_autograd():
    """tests/golden/make_golden_pairwise_grad.py: torch autograd through the reference's contact.py:23-47."""
    g = gio.load('pairwise_grad.npz')
    for tag, squared in (('sq', True), ('root', False)):
        for b in range(g['x'].shape[0]):
            gx, gy = oc.pairwise_adjoint(g['x'][b], g['y'][b], g['G'][b], squared)
            assert_close(gx, g['gx_' + tag][b], 1e-5, 2e-5, 'grad x ' + tag)
            assert_close(gy, g['gy_' + tag][b], 1e-5, 2e-5, 'grad y ' + tag)


def test_solid_angle_adjoint_vs_reference_autograd():
    """tests/golden/make_golden_solid_angle_grad.py: torch autograd through the reference's contact.py:49-147."""
    g = gio.load('solid_angle_grad.npz')
    for b in range(g['points'].shape[0]):
        gp, gt = oc.solid_angle_adjoint(g['points'][b], g['triangles'][b], g['G'][b])
        assert_close(gp, g['sa_grad_points'][b], 1e-4, 2e-4 * np.abs(g['sa_grad_points']).max(), 'd solid angles / d points')
        assert_close(gt, g['sa_grad_triangles'][b], 1e-4, 2e-4 * np.abs(g['sa_grad_triangles']).max(), 'd solid angles / d triangles')
        gw = np.repeat(g['gw'][b][:, None], g['triangles'].shape[1], 1) / (4 * np.pi)
        gp, gt = oc.solid_angle_adjoint(g['points'][b], g['triangles'][b], gw)
        assert_close(gp, g['w_grad_points'][b], 1e-4, 2e-4 * np.abs(g['w_grad_points']).max(), 'd winding / d points')
        assert_close(gt, g['w_grad_triangles'][b], 1e-4, 2e-4 * np.abs(g['w_grad_triangles']).max(), 'd winding / d triangles')


@pytest.mark.parametrize('tag', TAGS + FULL)
def test_v2v_min_masked(tag):
    g, gm = golden(tag), golden_mask(tag)
    for b in range(g['verts'].shape[0]):
        mn, arg = oc.v2v_min_masked(g['verts'][b], gm)
        assert_close(mn, g['v2v_min'][b], 0, 1e-6, 'v2v min')
        same = arg == g['v2v_argmin'][b]
        assert same.mean() > 0.99
        # where the argmin differs the two candidates must be tied within bmm noise
        v = g['verts'][b].astype(np.float64)
        d_ours = ((v - v[arg]) ** 2).sum(1)
        d_ref = ((v - v[g['v2v_argmin'][b]]) ** 2).sum(1)
        assert np.all(np.abs(d_ours - d_ref)[~same] < 2e-6)


def test_solid_angles_dense():
    g = golden('small')
    sa = oc.solid_angles(g['verts'][0], oc.gather_tris(g['verts'][0], g['faces']))
    assert_close(sa, g['solid_angles_b0'], 1e-5, 1e-5, 'solid angles')


@pytest.mark.parametrize('tag', TAGS + FULL)
def test_winding(tag):
    g = golden(tag)
    for b in range(g['verts'].shape[0]):
        w = oc.winding_numbers(g['verts'][b], oc.gather_tris(g['verts'][b], g['faces']))
        # float32 round-off of identical formulas; an ill-conditioned (query almost in the
        # plane of a nearby triangle) term may move a single vertex by a few 1e-5
        err = np.abs(w - g['winding'][b])
        assert np.percentile(err, 99.9) < 5e-6
        assert_close(w, g['winding'][b], 0, 2e-4, 'winding')
        clear = np.abs(g['winding'][b] - 0.99) > 1e-4
        assert np.array_equal((w <= 0.99)[clear], (g['winding'][b] <= 0.99)[clear])


@pytest.mark.parametrize('tag', TAGS + FULL)
def test_segments(tag):
    g = golden(tag)
    segs = oracle_segments(g)
    if 'segment_faces_flat' in g:
        for s, f in zip(segs, gio.unpack_ragged('segment_faces', g)):
            assert np.array_equal(s.faces.ravel(), f)
    for b in range(g['verts'].shape[0]):
        ext = np.concatenate([s.exterior(g['verts'][b]) for s in segs])
        want = g['segment_exterior'][b].astype(bool)
        assert (ext != want).sum() <= 1   # a flag may sit within 1e-6 of the 0.99 threshold


def _smplify_total(g, gm, segs, eucl):
    b_count, v_count = g['verts'].shape[:2]
    clw = float(g['contact_loss_weight'])
    has_dc = g['has_discrete_contact'] if 'has_discrete_contact' in g else np.ones(b_count, bool)
    ignore = g['ignore_idxs'] if 'ignore_idxs' in g else np.zeros(b_count, bool)
    total, grad = 0.0, np.zeros((b_count, v_count, 3))
    for b in range(b_count):
        if ignore[b]:                      # losses.py:74: bodies in ignore_idxs get no contact terms
            continue
        rp = region_pair_lists(g, b) if has_dc[b] else None
        r = oc.smplify_contact_body(g['verts'][b], g['faces'], gm, eucl, segs, rp)
        total += 10 * r['contact'] + clw * r['r2r']            # losses.py:120
        grad[b] = 10 * r['grad_contact'] + clw * r['grad_r2r']
    return total, grad


@pytest.mark.parametrize('tag', TAGS)
@pytest.mark.parametrize('eu', ['e0', 'e2'])
@pytest.mark.parametrize('sg', ['nos', 'seg'])
def test_smplify_contact_loss(tag, eu, sg):
    g, gm = golden(tag), golden_mask(tag)
    eucl = 0.0 if eu == 'e0' else float(g['euclthres'])
    total, grad = _smplify_total(g, gm, oracle_segments(g) if sg == 'seg' else None, eucl)
    key = 'smplify_%s_%s_contact' % (eu, sg)
    assert_close(total, g[key + '_loss'], 1e-5, 0, key)
    scale = np.abs(g[key + '_grad_verts']).max()
    assert_close(grad, g[key + '_grad_verts'], 1e-4, 1e-6 * scale, key + ' grad')


@pytest.mark.parametrize('tag', FULL)
@pytest.mark.parametrize('eu', ['e0', 'e2'])
def test_smplify_contact_loss_fullsize(eu, tag):
    g, gm = golden(tag), golden_mask(tag)
    eucl = 0.0 if eu == 'e0' else float(g['euclthres'])
    total, grad = _smplify_total(g, gm, oracle_segments(g), eucl)
    key = 'smplify_%s_seg_contact' % eu
    assert_close(total, g[key + '_loss'], 1e-5, 0, key)
    scale = np.abs(g[key + '_grad_verts']).max()
    assert_close(grad, g[key + '_grad_verts'], 1e-4, 1e-6 * scale, key + ' grad')


@pytest.mark.parametrize('tag', TAGS)
@pytest.mark.parametrize('use_hd', [False, True])
def test_train_contact_loss(tag, use_hd):
    g, gm = golden(tag), golden_mask(tag)
    loss, grad, _ = oc.train_contact_loss(
        g['verts'], g['valid_fit'], g['faces'], gm, float(g['euclthres']), oracle_segments(g),
        use_hd, hd_idx=g['hd_idx'], hd_w=g['hd_w'], hd_face=g['hd_face'])
    key = 'train_hd' if use_hd else 'train_plain'
    assert_close(loss, g[key + '_loss'], 1e-5, 0, key)
    scale = np.abs(g[key + '_grad_verts']).max()
    assert_close(grad, g[key + '_grad_verts'], 1e-4, 2e-6 * scale + TANH_QUANTUM, key + ' grad')


@pytest.mark.parametrize('tag', TAGS)
def test_contact_from_verts(tag):
    g = golden(tag)
    regions, pairs = gio.unpack_regions(g)
    pc = oc.contact_from_verts(g['verts'], regions, pairs)
    assert_close(pc, g['contact_from_verts'], 0, 1e-6, 'contact_from_verts')


def test_known_answers_tetrahedron():
    """Math KAT: winding number of a closed mesh is 1 inside, 0 outside."""
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    f = np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]], np.int64)
    pts = np.array([[0.1, 0.1, 0.1], [2, 2, 2], [0.2, 0.3, 0.1]], np.float32)
    w = oc.winding_numbers(pts, oc.gather_tris(v, f))
    assert_close(w, [1, 0, 1], 0, 1e-6, 'tetra')
    # on a vertex: incident faces contribute atan2(0,0)=0
    w0 = oc.winding_numbers(v[:1], oc.gather_tris(v, f))
    assert np.isfinite(w0).all()


def test_torch_chain_matches_reference_goldens():
    """The torch op-chain used as the CPU baseline computes what the reference computes."""
    import torch
    from oracle import torch_chain as tc
    g, gm = golden('medium'), golden_mask('medium')
    faces = torch.tensor(g['faces'])
    for b in range(g['verts'].shape[0]):
        v = torch.tensor(g['verts'][b])
        w = tc.winding(v[None], v[faces][None])[0].numpy()
        assert_close(w, g['winding'][b], 0, 2e-6, 'torch chain winding')
    p = tc.pairwise_sq(torch.tensor(g['verts'][:1]), torch.tensor(g['verts'][:1]))
    p[:, ~torch.tensor(gm)] = float('inf')
    mn, arg = torch.min(p, dim=1)
    assert_close(mn[0].numpy(), g['v2v_min'][0], 0, 1e-6, 'torch chain v2v')
    assert (arg[0].numpy() == g['v2v_argmin'][0]).mean() > 0.99
    val = tc.contact_forward_one_body(torch.tensor(g['verts'][0]), faces, torch.tensor(gm), 0.02)
    r = oc.smplify_contact_body(g['verts'][0], g['faces'], gm, 0.02, None, None)
    assert_close(val.item(), r['contact'], 1e-4, 1e-6, 'torch chain contact term')


@pytest.mark.parametrize('tag', TAGS)
def test_eft_contact_loss(tag):
    """EFT variant (tuch/eft/loss.py:129-181): means instead of sums, 100 * (contact + 0.5 r2r)."""
    g, gm = golden(tag), golden_mask(tag)
    segs = oracle_segments(g)
    for b in range(g['verts'].shape[0]):
        r = oc.eft_contact_body(g['verts'][b], g['faces'], gm, segs, region_pair_lists(g, b))
        total = 100 * (r['contact'] + 0.5 * r['r2r'])
        grad = 100 * (r['grad_contact'] + 0.5 * r['grad_r2r'])
        assert_close(total, g['eft_loss'][b], 1e-5, 1e-6, 'eft loss')
        scale = np.abs(g['eft_grad_verts'][b]).max()
        assert_close(grad, g['eft_grad_verts'][b], 1e-4, 2e-6 * scale, 'eft grad')


def _full_train(tag='full'):
    data = gio.load('contact_%s_train.npz' % tag)
    return {k: data[k] for k in data.files}


@pytest.mark.parametrize('tag', FULL)
@pytest.mark.parametrize('use_hd', [False, True])
def test_train_contact_loss_fullsize(use_hd, tag):
    """a7 at SMPL size (V=6890, N_hd=41328), plain and HD branch (loss.py:240-317)."""
    g, gm, gt = golden(tag), golden_mask(tag), _full_train(tag)
    loss, grad, _ = oc.train_contact_loss(
        g['verts'], np.ones(g['verts'].shape[0], bool), g['faces'], gm, float(g['euclthres']), oracle_segments(g),
        use_hd, hd_idx=g['hd_idx'], hd_w=g['hd_w'], hd_face=g['hd_face'])
    key = 'train_hd' if use_hd else 'train_plain'
    assert_close(loss, gt[key + '_loss'], 1e-5, 0, key)
    scale = np.abs(gt[key + '_grad_verts']).max()
    assert_close(grad, gt[key + '_grad_verts'], 1e-4, 2e-6 * scale, key + ' grad')


@pytest.mark.parametrize('tag', FULL)
def test_eft_contact_loss_fullsize(tag):
    g, gm, gt = golden(tag), golden_mask(tag), _full_train(tag)
    for b in range(g['verts'].shape[0]):
        r = oc.eft_contact_body(g['verts'][b], g['faces'], gm, oracle_segments(g), region_pair_lists(g, b))
        assert_close(100 * (r['contact'] + 0.5 * r['r2r']), gt['eft_loss'][b], 1e-5, 1e-6, 'eft loss')
        grad = 100 * (r['grad_contact'] + 0.5 * r['grad_r2r'])
        scale = np.abs(gt['eft_grad_verts'][b]).max()
        assert_close(grad, gt['eft_grad_verts'][b], 1e-4, 2e-6 * scale, 'eft grad')


@pytest.mark.parametrize('tag', TAGS)
def test_smplify_objective_terms(tag):
    """oracle/smplify.py against the reference's projection, gmof, GMM prior and the full stage-2 objective."""
    from oracle import smplify as osm
    g, gm = golden(tag), golden_mask(tag)
    proj = osm.perspective_projection(g['model_joints'], g['camera_t'], 5000., g['camera_center'])
    assert_close(proj, g['projected_joints'], 1e-6, 1e-4, 'projection')
    assert_close(osm.gmof(g['joints_2d'] - proj, 100.), g['gmof_values'], 1e-5, 1e-5, 'gmof')
    gmm = {k: g['gmm_' + k] for k in ('means', 'covars', 'weights')}
    assert_close(osm.merged_prior(g['body_pose'], gmm), g['prior_values'], 1e-5, 1e-4, 'prior')
    assert_close(osm.reprojection(g['model_joints'], g['camera_t'], g['camera_center'], g['joints_2d'],
                                  g['joints_conf']), g['body_fitting_reprojection'], 1e-5, 1e-4, 'reprojection')
    b_count = g['verts'].shape[0]
    rp = [region_pair_lists(g, b) if g['has_discrete_contact'][b] else None for b in range(b_count)]
    for eu, eucl in (('e0', 0.0), ('e2', float(g['euclthres']))):
        total, _, _ = osm.stage2_objective(
            g['verts'], g['model_joints'], g['body_pose'], g['faces'], gm, eucl, g['camera_t'], g['camera_center'],
            g['joints_2d'], g['joints_conf'], gmm, oracle_segments(g), rp, g['ignore_idxs'],
            contact_loss_weight=float(g['contact_loss_weight']))
        assert_close(total, g['smplify_%s_seg_full_loss' % eu], 1e-5, 0, 'stage-2 objective ' + eu)
