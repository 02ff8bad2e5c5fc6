"""GPU: the SMPLify-DC loop (a9 of SURVEY.md §8a).  The stage-2 gradient with respect to the
optimised parameters is checked against a gradient composed from the CPU oracle pieces
(oracle LBS autograd fed with the oracle's analytic contact gradient), and the optimiser is run
end to end on a small model."""
import numpy as np
import pytest
import torch

import golden_io as gio
from helpers import assert_close, close_logged, golden, golden_mask, oracle_segments, region_pair_lists
from oracle import contact as oc
from oracle import lbs as ol
from synthetic import make_body, random_poses

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
# loop-level tolerance (10 + 10 Adam iterations through float atomics): 3 x the observed maximum, see close_logged's log
LOOP_RTOL = 1e-5      # observed <= 1.6e-6 of the largest entry (profiles/r04_parity_counts.txt)


def _setup(batch, seed):
    from tuch_amd.models.smpl import SMPL
    from tuch_amd.smplify.prior import MaxMixturePrior
    from tuch_amd.utils.segmentation import BatchBodySegment
    body = make_body(14, 16, relax_iters=40)
    smpl = SMPL(model_data=body, batch_size=batch).to(DEV)
    prior = MaxMixturePrior(num_gaussians=8, gmm=body.gmm).to(DEV)
    bp, go, be = random_poses(batch, seed)
    rng = np.random.default_rng(seed)
    t = lambda a: torch.tensor(a, device=DEV)
    faces = t(body.faces)
    segments = BatchBodySegment(list(body.segments.keys()), faces, body.segments) if body.segments else None
    cdict = {'classes': [list(p) for p in body.region_pairs], 'csig': dict(body.regions)}
    gt = (rng.random((batch, len(body.region_pairs))) < 0.05).astype(np.float32)
    kp = np.concatenate([rng.standard_normal((batch, 49, 2)).astype(np.float32) * 30,
                         (0.5 + 0.5 * rng.random((batch, 49, 1))).astype(np.float32)], 2)
    return dict(body=body, smpl=smpl, prior=prior, bp=bp, go=go, be=be, segments=segments, cdict=cdict, gt=gt,
                kp=kp, cam_t=np.tile([[0., 0., 20.]], (batch, 1)).astype(np.float32), t=t)


def test_stage2_gradient_matches_oracle_composition():
    from tuch_amd.smplify.losses import contact_fitting_loss
    batch = 3
    s = _setup(batch, 11)
    body, t = s['body'], s['t']
    gm = body.geodesics > 0.3
    bp = t(s['bp']).requires_grad_(True)
    go = t(s['go']).requires_grad_(True)
    out = s['smpl'](global_orient=go, body_pose=bp, betas=t(s['be']))
    zero_prior = lambda pose, betas: torch.zeros(pose.shape[0], device=DEV)
    face_tensor = t(body.faces)[None].repeat(batch, 1, 1)
    loss = contact_fitting_loss(bp, go, None, None, t(s['be']), out.joints, t(gm), 0.02, t(s['cam_t']),
                                torch.zeros(batch, 2, device=DEV), t(s['kp'][:, :, :2]),
                                torch.zeros(batch, 49, device=DEV), zero_prior, s['cdict'],
                                [t(s['gt']), None], torch.zeros(batch, dtype=torch.bool, device=DEV),
                                torch.ones(batch, dtype=torch.bool, device=DEV), out.vertices,
                                face_tensor=face_tensor, contact_loss_weight=2000.0, segments=s['segments'])
    loss.backward()
    # oracle: forward LBS (fp64 autograd), contact gradient from the analytic oracle
    m64 = ol.model_tensors(body, torch.float64)
    t64 = lambda a: torch.tensor(a, dtype=torch.float64)
    bp64, go64 = t64(s['bp']).requires_grad_(True), t64(s['go']).requires_grad_(True)
    v64, _ = ol.smpl_forward(m64, t64(s['be']), bp64, go64)
    osegs = [oc.Segment(n, body.faces, sg['vidx'], list(sg['bands'].values())) for n, sg in body.segments.items()]
    names = list(body.regions.keys())
    total, gv = 0.0, np.zeros((batch, body.num_verts, 3))
    v32 = out.vertices.detach().cpu().numpy()
    for b in range(batch):
        rp = [(body.regions[a], body.regions[c]) for k, (a, c) in enumerate(body.region_pairs) if s['gt'][b, k] == 1]
        r = oc.smplify_contact_body(v32[b], body.faces, gm, 0.02, osegs or None, rp)
        total += 10 * r['contact'] + 2000.0 * r['r2r']
        gv[b] = 10 * r['grad_contact'] + 2000.0 * r['grad_r2r']
    (v64 * torch.tensor(gv)).sum().backward()
    assert_close(loss.item(), total, 1e-4, 2000 * 1e-6 * max(float(s['gt'].sum()), 1.0), 'stage-2 loss')
    # the contact terms are invariant under a rigid rotation about the root, so d/d(global_orient) is
    # analytically zero: both sides hold cancellation noise there -> one absolute scale for both
    scale = float(bp64.grad.abs().max())
    for got, want, name in ((bp.grad, bp64.grad, 'body_pose'), (go.grad, go64.grad, 'global_orient')):
        assert_close(got.cpu().numpy(), want.numpy(), 2e-3, 2e-4 * scale, 'grad ' + name)


def test_pose_prior_gradient_rides_with_the_body_model_within_one_backward_pass(monkeypatch):
    """The fused stage-2 tail leaves the pose prior's gradient with the body model's autograd node (one add launch less
    per step).  Same gradients as autograd's own sum; and a gradient left behind by a backward pass that never reached
    the body model's node (only camera_t asked for) does not leak into a later pass through that node."""
    from tuch_amd import ops
    from tuch_amd.smplify.losses import contact_fitting_loss
    batch = 3
    s = _setup(batch, 11)
    body, t = s['body'], s['t']
    gm = t(body.geodesics > 0.3)
    face_tensor = t(body.faces)[None].repeat(batch, 1, 1)

    def build():
        bp = t(s['bp']).requires_grad_(True)
        go = t(s['go']).requires_grad_(True)
        cam = t(s['cam_t']).requires_grad_(True)
        out = s['smpl'](global_orient=go, body_pose=bp, betas=t(s['be']))
        loss = contact_fitting_loss(bp, go, None, None, t(s['be']), out.joints, gm, 0.02, cam,
                                    torch.zeros(batch, 2, device=DEV), t(s['kp'][:, :, :2]), t(s['kp'][:, :, 2]),
                                    s['prior'], s['cdict'], [t(s['gt']), None],
                                    torch.zeros(batch, dtype=torch.bool, device=DEV),
                                    torch.ones(batch, dtype=torch.bool, device=DEV), out.vertices,
                                    face_tensor=face_tensor, contact_loss_weight=2000.0, segments=s['segments'])
        return bp, go, cam, out, loss
    bp, go, cam, out, loss = build()
    node = out.vertices.grad_fn
    assert getattr(node, 'pose_ref', None) is not None and node.pose_ref() is bp   # the body model's node knows its pose tensor
    g_new = torch.autograd.grad(loss, [bp, go, cam], retain_graph=True)
    assert node.pose_grad_extra is None                         # consumed
    # autograd's own sum: the side channel off
    monkeypatch.setattr(ops, '_graph_task_id', lambda: -1)
    bp2, go2, cam2, out2, loss2 = build()
    g_ref = torch.autograd.grad(loss2, [bp2, go2, cam2])
    monkeypatch.undo()
    scale = float(g_ref[0].abs().max())
    for a, b, name in zip(g_new, g_ref, ('body_pose', 'global_orient', 'camera_t')):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), 1e-5, 1e-6 * max(scale, float(b.abs().max())), 'grad ' + name)
    # a pass that stops in front of the body model leaves the prior's gradient with its node ...
    torch.autograd.grad(loss, [cam], retain_graph=True)
    assert node.pose_grad_extra is not None
    # ... and a later pass through the node alone must not pick it up
    g_alone = torch.autograd.grad(out.vertices.sum(), [bp], retain_graph=True)[0]
    bp3, go3, cam3, out3, loss3 = build()
    g_want = torch.autograd.grad(out3.vertices.sum(), [bp3])[0]
    assert torch.equal(g_alone, g_want)


@pytest.mark.parametrize('use_contact', [True, False])
def test_smplifydc_runs_and_reduces_the_objective(use_contact):
    from tuch_amd.smplify.smplifydc import SMPLifyDC
    batch = 4
    s = _setup(batch, 5)
    body, t = s['body'], s['t']
    geod = t(body.geodesics)
    fitter = SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=6, focal_length=5000., geodistssmpl=geod,
                       geothres=0.3, euclthres=0.02, device=torch.device(DEV), smpl=s['smpl'],
                       pose_prior=s['prior'])
    # keypoints = projection of a perturbed pose, so that there is something to fit
    from tuch_amd.utils.geometry import perspective_projection
    with torch.no_grad():
        tgt = s['smpl'](global_orient=t(s['go']), body_pose=t(s['bp']) + 0.1, betas=t(s['be']))
        j2d = perspective_projection(tgt.joints, torch.eye(3, device=DEV)[None].expand(batch, -1, -1),
                                     t(s['cam_t']), 5000., torch.zeros(batch, 2, device=DEV))
    kp = torch.cat([j2d, torch.ones(batch, 49, 1, device=DEV)], 2)
    init_pose = torch.cat([t(s['go']), t(s['bp'])], 1)
    before = fitter.get_fitting_loss(init_pose, t(s['be']), t(s['cam_t']), torch.zeros(batch, 2, device=DEV),
                                     kp.clone()).sum().item()
    res = fitter(init_pose, t(s['be']), t(s['cam_t']), torch.zeros(batch, 2, device=DEV), kp.clone(),
                 use_contact=use_contact, contactlist=s['cdict'], gt_contact=[t(s['gt']), None],
                 ignore_idxs=torch.zeros(batch, dtype=torch.bool, device=DEV),
                 has_discrete_contact=torch.ones(batch, dtype=torch.bool, device=DEV),
                 contact_loss_weight=1.0, segments=s['segments'])
    verts, joints, pose, betas, cam, reproj, optiverts = res
    assert verts.shape == (batch, body.num_verts, 3) and joints.shape == (batch, 49, 3)
    assert pose.shape == (batch, 72) and betas.shape == (batch, 10) and cam.shape == (batch, 3)
    assert reproj.shape == (batch, 49) and len(optiverts) == 6
    assert torch.isfinite(verts).all() and torch.isfinite(reproj).all()
    assert reproj.sum().item() < before                       # the fit moved toward the keypoints
    assert not torch.equal(pose, init_pose)


def test_graph_replayed_loops_match_eager_loops():
    """SMPLifyDC replays its optimisation loops as hipGraphs; the result must be the eager one
    (up to the last-ulp order dependence of the gradient scatter atomics)."""
    from tuch_amd.smplify.smplifydc import SMPLifyDC
    from tuch_amd.utils.geometry import perspective_projection
    batch = 3
    s = _setup(batch, 23)
    body, t = s['body'], s['t']
    with torch.no_grad():
        tgt = s['smpl'](global_orient=t(s['go']), body_pose=t(s['bp']) + 0.1, betas=t(s['be']))
        j2d = perspective_projection(tgt.joints, torch.eye(3, device=DEV)[None].expand(batch, -1, -1),
                                     t(s['cam_t']), 5000., torch.zeros(batch, 2, device=DEV))
    kp = torch.cat([j2d, torch.ones(batch, 49, 1, device=DEV)], 2)
    init_pose = torch.cat([t(s['go']), t(s['bp'])], 1)
    results = []
    for use_graph in (False, True):
        fitter = SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=8, focal_length=5000.,
                           geodistssmpl=t(body.geodesics), geothres=0.3, euclthres=0.02,
                           device=torch.device(DEV), smpl=s['smpl'], pose_prior=s['prior'], use_graph=use_graph)
        results.append(fitter(init_pose, t(s['be']), t(s['cam_t']), torch.zeros(batch, 2, device=DEV), kp.clone(),
                              use_contact=True, contactlist=s['cdict'], gt_contact=[t(s['gt']), None],
                              ignore_idxs=torch.zeros(batch, dtype=torch.bool, device=DEV),
                              has_discrete_contact=torch.ones(batch, dtype=torch.bool, device=DEV),
                              contact_loss_weight=1.0, segments=s['segments']))
    eager, graph = results
    assert len(eager[6]) == len(graph[6]) == 8
    for a, b, name in zip(eager[:6], graph[:6], ('verts', 'joints', 'pose', 'betas', 'cam', 'reproj')):
        assert_close(b.detach().cpu().numpy(), a.detach().cpu().numpy(), 2e-3, 2e-4, name)
    assert_close(graph[6][-1].cpu().numpy(), eager[6][-1].detach().cpu().numpy(), 2e-3, 2e-4, 'last optiverts')


def _loop_golden():
    data = gio.load('smplify_loop.npz')
    return {k: data[k] for k in data.files}


@pytest.mark.parametrize('use_contact', [True, False])
@pytest.mark.parametrize('use_graph', [False, True])
def test_smplifydc_matches_the_reference_loop(use_contact, use_graph, monkeypatch):
    """a9: the 7-tuple of the REFERENCE'S OWN SMPLifyDC.__call__ (tests/golden/make_golden_smplify.py: its two Adam
    loops, 10 + 10 iterations, ignored bodies, has_gt_keypoints, both branches) against ours, eager and replayed
    as hipGraphs.  1e-3 on everything; the per-iteration vertices of stage 2 as well."""
    from tuch_amd.models.smpl import SMPL
    from tuch_amd.smplify.prior import MaxMixturePrior
    from tuch_amd.smplify.smplifydc import SMPLifyDC
    from tuch_amd.utils.segmentation import BatchBodySegment
    monkeypatch.setenv('TUCH_GRAPH_STRICT', '1')
    g = _loop_golden()
    body = make_body(int(g['rings']), int(g['segs']), relax_iters=int(g['relax_iters']))
    t = lambda a: torch.tensor(a, device=DEV)
    batch = g['init_pose'].shape[0]
    smpl = SMPL(model_data=body, batch_size=batch).to(DEV)
    prior = MaxMixturePrior(num_gaussians=8, gmm=body.gmm).to(DEV)
    segments = BatchBodySegment(list(body.segments.keys()), t(body.faces), body.segments)
    cdict = {'classes': [list(p) for p in body.region_pairs], 'csig': dict(body.regions)}
    fitter = SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=int(g['num_iters']), focal_length=5000.,
                       geodistssmpl=t(body.geodesics), geothres=float(g['geothres']), euclthres=float(g['euclthres']),
                       device=torch.device(DEV), smpl=smpl, pose_prior=prior, use_graph=use_graph)
    assert fitter.ign_joints == [1, 9, 12, 27, 28]
    kp = t(g['keypoints_2d'])
    res = fitter(t(g['init_pose']), t(g['init_betas']), t(g['init_cam_t']), t(g['camera_center']), kp,
                 use_contact=use_contact, contactlist=cdict, gt_contact=[t(g['gt_contact']), None],
                 ignore_idxs=t(g['ignore_idxs']), has_discrete_contact=t(g['has_discrete_contact']),
                 has_gt_keypoints=t(g['has_gt_keypoints']), contact_loss_weight=float(g['contact_loss_weight']),
                 segments=segments)
    if use_graph:
        assert fitter.graph_replayed == {'stage1': 7, 'stage2': 7}
    tag = 'contact' if use_contact else 'plain'
    names = ('vertices', 'joints', 'pose', 'betas', 'camera_translation', 'reprojection_loss')
    for name, got in zip(names, res[:6]):
        want = g['%s_%s' % (tag, name)]
        close_logged(got.detach().cpu().numpy(), want, LOOP_RTOL, LOOP_RTOL * np.abs(want).max(),
                     'SMPLifyDC loop %s %s graph=%d' % (tag, name, use_graph))
    optiverts = torch.stack([v.detach() for v in res[6]]).cpu().numpy()
    want = g['%s_optiverts' % tag]
    assert optiverts.shape == want.shape
    close_logged(optiverts, want, LOOP_RTOL, LOOP_RTOL, 'SMPLifyDC loop %s optiverts graph=%d' % (tag, use_graph))
    assert torch.equal(kp, t(g['keypoints_2d']))                 # __call__ does not write into its input
    # the fit actually moved: not a comparison of two untouched initial states
    assert np.abs(g['%s_pose' % tag] - g['init_pose']).max() > 0.05
    got = fitter.get_fitting_loss(t(g['init_pose']), t(g['init_betas']), t(g['init_cam_t']), t(g['camera_center']),
                                  kp.clone(), t(g['has_gt_keypoints']))
    assert_close(got.cpu().numpy(), g['get_fitting_loss'], 1e-4, 1e-3, 'get_fitting_loss')


def test_smplifydc_default_contactlist():
    """__call__ defaults contactlist=[] (smplifydc.py:70); with use_contact that means no region term."""
    from tuch_amd.smplify.smplifydc import SMPLifyDC
    batch = 2
    s = _setup(batch, 7)
    t = s['t']
    fitter = SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=3, focal_length=5000.,
                       geodistssmpl=t(s['body'].geodesics), geothres=0.3, euclthres=0.02, device=torch.device(DEV),
                       smpl=s['smpl'], pose_prior=s['prior'])
    init_pose = torch.cat([t(s['go']), t(s['bp'])], 1)
    res = fitter(init_pose, t(s['be']), t(s['cam_t']), torch.zeros(batch, 2, device=DEV), t(s['kp']),
                 use_contact=True, gt_contact=[t(s['gt']), None],
                 ignore_idxs=torch.zeros(batch, dtype=torch.bool, device=DEV),
                 has_discrete_contact=torch.ones(batch, dtype=torch.bool, device=DEV))
    assert torch.isfinite(res[0]).all() and len(res[6]) == 3


def test_config3_fit_batch32_fullsize(monkeypatch):
    """BASELINE configs[2]: demo_smplify_dc.py-style fit, batch 32, V=6890, 100 + 100 iterations, loops replayed as
    hipGraphs.  Checks: everything finite, the stage-2 objective decreases, and the objective the GPU reports at
    iterations 0, 1 and 99 equals the CPU oracle's evaluated at the recorded parameters (oracle LBS -> oracle
    contact / reprojection / prior), for a sample of the bodies at 0 and 99 and for all of them at iteration 1."""
    import bench
    from oracle import smplify as osm
    from tuch_amd.smplify.smplifydc import SMPLifyDC
    monkeypatch.setenv('TUCH_GRAPH_STRICT', '1')
    batch, iters = 32, 100
    dev = torch.device(DEV)
    p = bench.build_problem(batch, dev, 1003)
    body = p['body']
    fitter = SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=iters, focal_length=5000.,
                       geodistssmpl=torch.tensor(body.geodesics, device=dev), geothres=0.3, euclthres=0.02, device=dev,
                       smpl=p['smpl'], pose_prior=p['prior'], record_history=True)
    kp = torch.cat([p['j2d'], p['conf'][..., None]], 2)
    init_pose = torch.cat([p['global_orient'], p['body_pose']], 1)
    res = fitter(init_pose, p['betas'], p['cam_t'], p['cam_c'], kp, use_contact=True, contactlist=p['cdict'],
                 gt_contact=[p['gt'], None], ignore_idxs=p['ignore'], has_discrete_contact=p['has_dc'],
                 contact_loss_weight=2000.0, segments=p['segments'])
    verts, joints, pose, betas, cam, reproj, optiverts = res
    assert fitter.graph_replayed == {'stage1': iters - 3, 'stage2': iters - 3}
    assert len(optiverts) == iters and len(fitter.history['stage2']) == iters
    for x in (verts, joints, pose, betas, cam, reproj):
        assert torch.isfinite(x).all()
    losses = torch.stack([h['loss'] for h in fitter.history['stage2']]).cpu().numpy()
    assert np.isfinite(losses).all()
    assert losses[-1] < losses[0]
    # Adam at lr 1e-2 is not strictly monotone step by step; averaged over windows of 10 it is
    win = losses.reshape(10, 10).mean(1)
    assert np.all(np.diff(win) < 0.02 * np.abs(win[:-1])), win
    # ---- oracle evaluation at the recorded parameters
    m = ol.model_tensors(body)
    gm = body.geodesics > 0.3
    osegs = [oc.Segment(n, body.faces, sg['vidx'], list(sg['bands'].values())) for n, sg in body.segments.items()]
    gt = p['gt'].cpu().numpy()
    conf = kp[:, :, 2].clone()
    conf[:, fitter.ign_joints] = 0.0
    for it, sample in ((0, [0, 13, 31]), (1, list(range(batch))), (99, [0, 13, 31])):
        h = fitter.history['stage2'][it]
        bp, go = h['params'][0].cpu(), h['params'][1].cpu()
        v_o, j_o = ol.smpl_forward(m, betas.cpu(), bp, go)
        v_gpu = optiverts[it].cpu().numpy()
        close_logged(v_gpu, v_o.numpy(), 1e-4, 1e-5, 'config 3: vertices at iteration %d' % it)
        idx = np.asarray(sample)
        rp = [[(body.regions[a], body.regions[c]) for k, (a, c) in enumerate(body.region_pairs) if gt[b, k] == 1]
              for b in idx]
        total, per_body, parts = osm.stage2_objective(
            v_gpu[idx], j_o.numpy()[idx], bp.numpy()[idx], body.faces, gm, 0.02, cam.cpu().numpy()[idx],
            p['cam_c'].cpu().numpy()[idx], p['j2d'].cpu().numpy()[idx], conf.cpu().numpy()[idx], body.gmm, osegs, rp,
            None, contact_loss_weight=2000.0)
        if len(sample) == batch:
            n_sel = float(gt.sum())
            close_logged(float(losses[it]), total, 1e-4, 2000 * 1e-6 * n_sel, 'config 3: objective at iteration %d' % it)
        else:
            # the GPU reports the batch sum; compare the sampled bodies through a sub-batch evaluation on the GPU
            from tuch_amd.smplify.losses import contact_fitting_loss
            sel = torch.as_tensor(idx, device=dev)
            with torch.no_grad():
                out = p['smpl'](global_orient=h['params'][1][sel], body_pose=h['params'][0][sel], betas=betas[sel])
                sub = contact_fitting_loss(h['params'][0][sel], h['params'][1][sel], None, None, betas[sel], out.joints,
                                           fitter.geomask, 0.02, cam[sel], p['cam_c'][sel], p['j2d'][sel].contiguous(),
                                           conf[sel], p['prior'], cdict=p['cdict'], gt_contact=[p['gt'][sel], None],
                                           ignore_idxs=p['ignore'][sel], has_discrete_contact=p['has_dc'][sel],
                                           verts=out.vertices, face_tensor=fitter.face_tensor[:len(idx)],
                                           focal_length=5000., contact_loss_weight=2000.0, segments=p['segments'])
            n_sel = float(gt[idx].sum())
            close_logged(sub.item(), total, 1e-4, 2000 * 1e-6 * n_sel, 'config 3: objective of the sample at iteration %d' % it)


def test_loops_kept_between_calls_reproduce_fresh_fits(monkeypatch):
    """SMPLifyDC keeps its captured loops between calls (a training step runs 10 + 10 iterations per call): a second
    call with other inputs must give what a fitter without that cache gives, and calls must not disturb each other's
    returned tensors."""
    from tuch_amd.smplify.smplifydc import SMPLifyDC
    from tuch_amd.utils.geometry import perspective_projection
    monkeypatch.setenv('TUCH_GRAPH_STRICT', '1')
    batch = 3
    s = _setup(batch, 31)
    body, t = s['body'], s['t']

    def inputs(seed):
        rng = np.random.default_rng(seed)
        bp, go = s['bp'] + 0.05 * rng.standard_normal(s['bp'].shape).astype(np.float32), s['go']
        with torch.no_grad():
            tgt = s['smpl'](global_orient=t(go), body_pose=t(bp) + 0.1, betas=t(s['be']))
            j2d = perspective_projection(tgt.joints, torch.eye(3, device=DEV)[None].expand(batch, -1, -1),
                                         t(s['cam_t']), 5000., torch.zeros(batch, 2, device=DEV))
        kp = torch.cat([j2d, torch.tensor(rng.uniform(0.5, 1, (batch, 49, 1)).astype(np.float32), device=DEV)], 2)
        gt = t((rng.random((batch, len(body.region_pairs))) < 0.05).astype(np.float32))
        return torch.cat([t(go), t(bp)], 1), kp, gt

    def fit(fitter, pose, kp, gt):
        return fitter(pose, t(s['be']), t(s['cam_t']), torch.zeros(batch, 2, device=DEV), kp, use_contact=True,
                      contactlist=s['cdict'], gt_contact=[gt, None],
                      ignore_idxs=torch.tensor([False, True, False], device=DEV),
                      has_discrete_contact=torch.ones(batch, dtype=torch.bool, device=DEV),
                      has_gt_keypoints=torch.tensor([True, False, False], device=DEV),
                      contact_loss_weight=2000.0, segments=s['segments'])

    mk = lambda: SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=8, focal_length=5000., geodistssmpl=t(body.geodesics),
                           geothres=0.3, euclthres=0.02, device=torch.device(DEV), smpl=s['smpl'], pose_prior=s['prior'])
    cached = mk()
    a1 = fit(cached, *inputs(1))
    keep = [x.clone() for x in a1[:6]]
    a2 = fit(cached, *inputs(2))
    assert len(cached._sessions) == 1 and cached.graph_replayed == {'stage1': 8, 'stage2': 8}
    for x, k in zip(a1[:6], keep):
        assert torch.equal(x, k)                         # the first call's results were not overwritten by the second
    monkeypatch.setenv('TUCH_SMPLIFY_SESSIONS', '0')
    fresh = mk()
    b2 = fit(fresh, *inputs(2))
    assert len(fresh._sessions) == 0
    for a, b, name in zip(a2[:6], b2[:6], ('verts', 'joints', 'pose', 'betas', 'cam', 'reproj')):
        assert_close(a.cpu().numpy(), b.detach().cpu().numpy(), 2e-3, 2e-4, name)
    assert len(a2[6]) == len(b2[6]) == 8
    assert_close(a2[6][-1].cpu().numpy(), b2[6][-1].detach().cpu().numpy(), 2e-3, 2e-4, 'last optiverts')


def test_deterministic_mode_reproduces_a_fit_bit_for_bit():
    """Deterministic mode (the default; TUCH_DETERMINISTIC=0 / ops.set_deterministic(False) opt out): the gradient scatters of the stage-2 tail and of the SMPL
    adjoint accumulate 64-bit fixed-point numbers with integer atomics instead of float atomics -- the same fit twice,
    from fresh fitters, gives identical BITS in every output (vertices, joints, pose, betas, camera, reprojection loss,
    the per-iteration vertices); and the deterministic fit agrees with the float-atomic one to float tolerance."""
    from tuch_amd import ops
    from tuch_amd.smplify.smplifydc import SMPLifyDC
    batch = 4
    s = _setup(batch, 17)
    body, t = s['body'], s['t']

    def fit():
        fitter = SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=12, focal_length=5000., geodistssmpl=t(body.geodesics),
                           geothres=0.3, euclthres=0.02, device=torch.device(DEV), smpl=s['smpl'], pose_prior=s['prior'])
        out = fitter(torch.cat([t(s['go']), t(s['bp'])], 1), t(s['be']), t(s['cam_t']), torch.zeros(batch, 2, device=DEV),
                     t(s['kp']), use_contact=True, contactlist=s['cdict'], gt_contact=[t(s['gt']), None],
                     ignore_idxs=torch.zeros(batch, dtype=torch.bool, device=DEV),
                     has_discrete_contact=torch.ones(batch, dtype=torch.bool, device=DEV),
                     contact_loss_weight=2000.0, segments=s['segments'])
        torch.cuda.synchronize()
        return [x.detach().clone() for x in out[:6]] + [v.detach().clone() for v in out[6]]
    assert ops.deterministic()                         # the default since round 6
    with ops.deterministic_mode(False):
        plain = fit()
    first, second = fit(), fit()
    assert len(first) == len(second) > 6
    for a, b in zip(first, second):
        assert torch.equal(a, b)
    for a, b in zip(first[:6], plain[:6]):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), 2e-3, 2e-4 * max(float(b.abs().max()), 1e-3), 'deterministic vs float atomics')


def test_adam_inside_the_backward_kernel_matches_the_separate_launch(monkeypatch):
    """optim.Adam(fuse_backward=True): the body model's last backward kernel applies torch.optim.Adam's update to the two
    pose tensors itself (tuch_smpl_backward_split_adam), the optimiser's own launch is gone.  The same arithmetic in the
    same order per element: in deterministic mode (bit-reproducible gradients) a 12 + 12-iteration fit gives identical
    BITS with the update inside the kernel and with the separate tuch_adam_step launch; the kept stage-2 loop really took
    the fused path (its optimiser never launched a step of its own)."""
    from tuch_amd import ops, optim
    from tuch_amd.smplify.smplifydc import SMPLifyDC
    batch = 4
    s = _setup(batch, 29)
    body, t = s['body'], s['t']
    launches = []
    real = optim.Adam.step

    def counting(self):
        launches.append(not self._applied)        # True: this call launches tuch_adam_step itself
        return real(self)
    monkeypatch.setattr(optim.Adam, 'step', counting)

    def fit(fused):
        fitter = SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=12, focal_length=5000., geodistssmpl=t(body.geodesics),
                           geothres=0.3, euclthres=0.02, device=torch.device(DEV), smpl=s['smpl'], pose_prior=s['prior'])
        fitter.fused_adam = fused          # read when the kept stage-2 loop is built (first call)
        assert fitter.keep_sessions
        del launches[:]
        out = fitter(torch.cat([t(s['go']), t(s['bp'])], 1), t(s['be']), t(s['cam_t']), torch.zeros(batch, 2, device=DEV),
                     t(s['kp']), use_contact=True, contactlist=s['cdict'], gt_contact=[t(s['gt']), None],
                     ignore_idxs=torch.zeros(batch, dtype=torch.bool, device=DEV),
                     has_discrete_contact=torch.ones(batch, dtype=torch.bool, device=DEV),
                     contact_loss_weight=2000.0, segments=s['segments'])
        torch.cuda.synchronize()
        return [x.detach().clone() for x in out[:6]] + [v.detach().clone() for v in out[6]], list(launches)
    with ops.deterministic_mode(True), ops.off_default_stream(DEV):
        fused, steps_fused = fit(True)
        plain, steps_plain = fit(False)
    for a, b in zip(fused, plain):
        assert torch.equal(a, b)
    assert float((fused[2] - torch.cat([t(s['go']), t(s['bp'])], 1)).abs().max()) > 0.02        # the fit moved
    # step() is called in the three eager iterations and the capture of each kept loop: stage 1 always launches its own
    # update, stage 2 only without the fusion
    assert sum(steps_plain) > sum(steps_fused) >= 1 and sum(steps_plain) - sum(steps_fused) >= 3, (steps_fused, steps_plain)


def test_fused_adam_only_when_the_stage2_objective_is_the_root_of_the_pass():
    """optim.Adam(fuse_backward=True) may apply its update inside the body model's backward kernel only when EVERY gradient
    of the two pose tensors flows through the stage-2 objective node.  `tail + extra(body_pose)` sent through
    ops.backward_scalar reaches that node with the same cached unit seed (AddBackward forwards it unchanged): the update must
    then be left to step(), with the extra term's gradient in it.  Also: a fused update whose step() never came does not
    swallow the next one (zero_grad clears the flag), and a second backward before step() does not move the parameters twice."""
    from tuch_amd import ops, optim
    from tuch_amd.smplify.losses import contact_fitting_loss
    batch = 3
    s = _setup(batch, 31)
    body, t = s['body'], s['t']
    gm = t(body.geodesics > 0.3)
    face_tensor = t(body.faces)[None].repeat(batch, 1, 1)

    def objective(bp, go):
        out = s['smpl'](global_orient=go, body_pose=bp, betas=t(s['be']))
        return contact_fitting_loss(bp, go, None, None, t(s['be']), out.joints, gm, 0.02, t(s['cam_t']),
                                    torch.zeros(batch, 2, device=DEV), t(s['kp'][:, :, :2]), t(s['kp'][:, :, 2]),
                                    s['prior'], s['cdict'], [t(s['gt']), None],
                                    torch.zeros(batch, dtype=torch.bool, device=DEV),
                                    torch.ones(batch, dtype=torch.bool, device=DEV), out.vertices,
                                    face_tensor=face_tensor, contact_loss_weight=2000.0, segments=s['segments'])

    def one_step(fused, with_extra):
        bp = t(s['bp']).requires_grad_(True)
        go = t(s['go']).requires_grad_(True)
        opt = optim.make_adam([bp, go], 1e-2, fuse_backward=fused)
        assert isinstance(opt, optim.Adam)
        loss = objective(bp, go)
        if with_extra:
            loss = loss + 1e4 * (bp ** 2).sum()
        opt.zero_grad(set_to_none=True)
        ops.backward_scalar(loss)
        applied = opt._applied
        opt.step()
        torch.cuda.synchronize()
        return bp.detach().clone(), go.detach().clone(), applied
    before = ops.deterministic()
    ops.set_deterministic(True)
    try:
        with ops.off_default_stream(DEV):
            bp_f, go_f, applied_f = one_step(True, False)
            bp_p, go_p, applied_p = one_step(False, False)
            assert applied_f and not applied_p                       # the plain objective IS the root: fused
            assert torch.equal(bp_f, bp_p) and torch.equal(go_f, go_p)
            bp_fx, go_fx, applied_fx = one_step(True, True)
            bp_px, go_px, _ = one_step(False, True)
            assert not applied_fx                                    # not the root: the update waits for step()
            assert torch.equal(bp_fx, bp_px) and torch.equal(go_fx, go_px)
            assert float((bp_fx - bp_f).abs().max()) > 1e-3          # ... and the extra term's gradient is in it
            # a fused update without its step(): the next zero_grad() / step() pair is a real step again
            bp = t(s['bp']).requires_grad_(True)
            go = t(s['go']).requires_grad_(True)
            opt = optim.make_adam([bp, go], 1e-2, fuse_backward=True)
            ops.backward_scalar(objective(bp, go))
            assert opt._applied
            moved = bp.detach().clone()
            ops.backward_scalar(objective(bp, go))                   # second pass before step(): gradients only
            torch.cuda.synchronize()
            assert torch.equal(bp.detach(), moved)
            opt.zero_grad(set_to_none=True)
            assert not opt._applied
            (bp ** 2).sum().backward()
            go.grad = torch.zeros_like(go)
            opt.step()
            torch.cuda.synchronize()
            assert not torch.equal(bp.detach(), moved)               # this step() was not skipped
    finally:
        ops.set_deterministic(before)


@pytest.mark.parametrize('shape_w', [1.0, 0.0])
def test_stage1_objective_in_one_launch_vs_reference_and_torch_ops(shape_w):
    """camera_fitting_loss (losses.py:125-152) on a HIP device is ONE kernel (ops._Stage1Objective): the value against the
    reference's own golden (shape_prior_weight = 1) and, value + gradients w.r.t. joints / camera translation / betas,
    against the torch-op form of the same function (autograd), also under a non-unit upstream gradient."""
    import types
    from tuch_amd.smplify import losses
    g = golden('medium')
    t = lambda a: torch.tensor(a, device=DEV)

    def run(fused, scale):
        losses.STAGE1_FUSED = fused
        try:
            j = t(g['model_joints']).requires_grad_(True)
            cam = t(g['camera_t']).requires_grad_(True)
            be = t(g['betas']).requires_grad_(True)
            out = types.SimpleNamespace(joints=j, betas=be)
            val = losses.camera_fitting_loss(out, cam, t(g['camera_t_est']), t(g['camera_center']), t(g['joints_2d']),
                                             t(g['joints_conf']), focal_length=5000., shape_prior_weight=shape_w)
            (val * scale).backward()
            return val.item(), j.grad.cpu().numpy(), cam.grad.cpu().numpy(), None if be.grad is None else be.grad.cpu().numpy()
        finally:
            losses.STAGE1_FUSED = True
    for scale in (1.0, 0.37):
        v1, gj1, gc1, gb1 = run(True, scale)
        v0, gj0, gc0, gb0 = run(False, scale)
        if shape_w == 1.0:
            assert_close(v1, g['camera_fitting_loss'], 1e-5, 0, 'fused stage-1 objective vs the reference')
        assert_close(v1, v0, 1e-5, 0, 'value')
        for a, b_, name in ((gj1, gj0, 'joints'), (gc1, gc0, 'camera_t')):
            assert_close(a, b_, 1e-4, 1e-6 * np.abs(b_).max(), 'grad ' + name)
        if shape_w:
            assert_close(gb1, gb0, 1e-5, 1e-7, 'grad betas')
        else:
            assert gb1 is None or not gb1.any()


def test_fixed_point_vertex_gradient_reaches_every_kind_of_backward_pass():
    """Deterministic mode (the default): the stage-2 node leaves its vertex gradient in 64-bit fixed-point accumulators and
    the body model's skinning adjoint reads them there -- no conversion launch in a fit's `objective.backward()`.  Every
    other way of asking gets the same numbers: loss.backward() (not through ops.backward_scalar), a scaled loss, the
    gradient of the vertices themselves (torch.autograd.grad / retain_grad), a loss with a further term."""
    from tuch_amd import ops
    from tuch_amd.smplify.losses import contact_fitting_loss
    batch = 3
    s = _setup(batch, 41)
    body, t = s['body'], s['t']
    gm = t(body.geodesics > 0.3)
    face_tensor = t(body.faces)[None].repeat(batch, 1, 1)
    assert ops.deterministic()

    def graph(retain=False):
        bp = t(s['bp']).requires_grad_(True)
        go = t(s['go']).requires_grad_(True)
        out = s['smpl'](global_orient=go, body_pose=bp, betas=t(s['be']))
        if retain:
            out.vertices.retain_grad()
        loss = contact_fitting_loss(bp, go, None, None, t(s['be']), out.joints, gm, 0.02, t(s['cam_t']),
                                    torch.zeros(batch, 2, device=DEV), t(s['kp'][:, :, :2]), t(s['kp'][:, :, 2]),
                                    s['prior'], s['cdict'], [t(s['gt']), None],
                                    torch.zeros(batch, dtype=torch.bool, device=DEV),
                                    torch.ones(batch, dtype=torch.bool, device=DEV), out.vertices,
                                    face_tensor=face_tensor, contact_loss_weight=2000.0, segments=s['segments'])
        return bp, go, out, loss
    with ops.off_default_stream(DEV):
        bp, go, _, loss = graph()
        ops.backward_scalar(loss)                                   # the fit's pass: accumulators handed to the body model
        want_bp, want_go = bp.grad.clone(), go.grad.clone()
        assert float(want_bp.abs().max()) > 0
        bp, go, _, loss = graph()
        loss.backward()                                             # a plain backward: converted, the same bits
        assert torch.equal(bp.grad, want_bp) and torch.equal(go.grad, want_go)
        bp, go, _, loss = graph()
        ops.backward_scalar(loss)                                   # ... and twice the same (bit-reproducible)
        assert torch.equal(bp.grad, want_bp) and torch.equal(go.grad, want_go)
        bp, go, _, loss = graph()
        (3.0 * loss).backward()                                     # a scaled upstream gradient
        assert_close(bp.grad.cpu().numpy(), 3.0 * want_bp.cpu().numpy(), 1e-5, 1e-6 * float(want_bp.abs().max()), 'scaled')
        # the vertices' own gradient: real numbers, not the carrier of zeros -- and they give the same pose gradient
        bp, go, out, loss = graph()
        gv, = torch.autograd.grad(loss, out.vertices, retain_graph=True)
        assert float(gv.abs().max()) > 0
        bp2, go2, out2, loss2 = graph(retain=True)
        ops.backward_scalar(loss2)
        assert torch.equal(out2.vertices.grad, gv)
        assert torch.equal(bp2.grad, want_bp) and torch.equal(go2.grad, want_go)
        with ops.deterministic_mode(False):
            bp3, go3, out3, loss3 = graph(retain=True)
            ops.backward_scalar(loss3)
        assert_close(out3.vertices.grad.cpu().numpy(), gv.cpu().numpy(), 1e-4, 1e-6 * float(gv.abs().max()), 'fixed vs float atomics')
        # a further term on the vertices: autograd adds its gradient to the carrier, the adjoint adds the accumulators
        bp, go, out, loss = graph()
        extra = (out.vertices ** 2).sum()
        ops.backward_scalar(loss + extra)
        bp4, go4, out4, _ = graph()
        (out4.vertices ** 2).sum().backward()
        assert_close(bp.grad.cpu().numpy(), (want_bp + bp4.grad).cpu().numpy(), 1e-5, 1e-6 * float(want_bp.abs().max()), 'with a further term')
