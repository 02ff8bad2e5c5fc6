"""GPU: the callers of the hot path (BASELINE configs 4/5) against goldens produced by the REFERENCE'S OWN
``TUCH.forward_train_step`` and ``FitsDict`` (tests/golden/make_golden_train.py): one training step with SMPLify-DC in
the loop, the dictionary of best fits on the device, the valid-fit logic and RegressorLoss (HD branch)."""
import types

import numpy as np
import pytest
import torch

import golden_io as gio
from helpers import assert_close, close_logged
from synthetic import make_body, make_regressor

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
# step-level tolerances (5 + 5 Adam iterations inside the step, float atomics): 3 x the observed maxima (close_logged's log)
STEP_RTOL = 1e-5           # observed <= 9.3e-7
STEP_GRAD_RTOL = 5e-5      # observed 1.2e-6 of the largest entry


def _golden():
    data = gio.load('train_step.npz')
    return {k: data[k] for k in data.files}


def _options(g, tmp):
    o = types.SimpleNamespace(checkpoint_dir=str(tmp))
    for k in g:
        if k.startswith('opt_'):
            v = g[k]
            setattr(o, k[4:], v.item() if v.shape == () else v)
    return o


def _datasets(g):
    names = [k[len('static_fits_'):] for k in g if k.startswith('static_fits_')]
    return types.SimpleNamespace(dataset_dict={n: i for i, n in enumerate(names)},
                                 datasets=[list(range(len(g['static_fits_' + n]))) for n in names]), names


def _batch(g):
    out = {}
    for k in g:
        if k.startswith('batch_'):
            v = g[k]
            out[k[6:]] = [str(x) for x in v] if v.dtype.kind in 'US' else torch.tensor(v, device=DEV)
    return out


def test_fits_dict_gather_and_scatter_match_the_reference(tmp_path):
    from tuch_amd.train.fits_dict import FitsDict
    g = _golden()
    train_ds, names = _datasets(g)
    for n in names:
        np.save(tmp_path / (n + '_fits.npy'), g['static_fits_' + n])
    fd = FitsDict(_options(g, tmp_path), train_ds, device=DEV)
    b = _batch(g)
    key = (b['dataset_name'], b['sample_index'], b['rot_angle'], b['is_flipped'])
    pose, betas = fd[key]
    assert pose.device.type == 'cuda'
    assert_close(pose.cpu().numpy(), g['fits_get_pose'], 1e-5, 2e-5, 'fits[...] pose')
    assert_close(betas.cpu().numpy(), g['fits_get_betas'], 0, 0, 'fits[...] betas')
    fd[key + (torch.tensor(g['fits_set_update'], device=DEV),)] = (torch.tensor(g['fits_set_pose'], device=DEV), betas + 0.1)
    for n in names:
        assert_close(fd.fits_dict[n].cpu().numpy(), g['fits_after_set_' + n], 1e-5, 2e-5, 'table ' + n)
    fd.save()
    assert np.allclose(np.load(tmp_path / (names[0] + '_fits.npy')), fd.fits_dict[names[0]].cpu().numpy())


def test_fits_dict_duplicate_rows_follow_the_reference_order(tmp_path):
    """A sample index that occurs twice in one batch (MixedDataset wraps small datasets): the reference writes sample by
    sample, only where ``update`` is set (fits_dict.py:75-85) -- the last updating occurrence wins, a non-updating one
    never overwrites.  All four orders of (update, no update) on duplicated rows, on two datasets."""
    from tuch_amd.train.fits_dict import FitsDict
    rng = np.random.default_rng(8)
    sizes = {'dsA': 7, 'dsB': 5}
    for n, k in sizes.items():
        np.save(tmp_path / (n + '_fits.npy'), rng.standard_normal((k, 82)).astype(np.float32))
    ds = types.SimpleNamespace(dataset_dict={n: i for i, n in enumerate(sizes)}, datasets=[list(range(k)) for k in sizes.values()])
    fd = FitsDict(types.SimpleNamespace(checkpoint_dir=str(tmp_path)), ds, device=DEV)
    before = {n: t.cpu().numpy().copy() for n, t in fd.fits_dict.items()}
    names = ['dsA', 'dsB', 'dsA', 'dsA', 'dsB', 'dsA', 'dsB', 'dsA', 'dsA', 'dsB', 'dsA', 'dsA']
    index = [3, 1, 3, 0, 1, 0, 4, 5, 5, 2, 6, 6]
    update = [1, 0, 0, 0, 1, 1, 1, 1, 1, 0, 0, 0]      # A3: (1,0)  B1: (0,1)  A0: (0,1)  A5: (1,1)  A6: (0,0)
    pose = (0.4 * rng.standard_normal((12, 72))).astype(np.float32)     # |global orient| < pi: axis-angle round trip is the identity
    betas = rng.standard_normal((12, 10)).astype(np.float32)
    z = torch.zeros(12, device=DEV)
    for _ in range(2):                                  # twice: the scratch column must be clean again after a call
        fd[(names, torch.tensor(index, device=DEV), z, z, torch.tensor(update, device=DEV))] = \
            (torch.tensor(pose, device=DEV), torch.tensor(betas, device=DEV))
    want = {n: t.copy() for n, t in before.items()}
    for n, (name, i, u) in enumerate(zip(names, index, update)):
        if u:
            want[name][i] = np.concatenate([pose[n], betas[n]])
    for n in sizes:
        got = fd.fits_dict[n].cpu().numpy()
        assert_close(got, want[n], 1e-5, 1e-6, 'table ' + n)       # rot = 0: the round trip through rotation matrices
        assert np.array_equal(got[[1, 2]] if n == 'dsA' else got[[0, 3]], before[n][[1, 2]] if n == 'dsA' else before[n][[0, 3]])


def test_forward_train_step_matches_the_reference(tmp_path):
    """run_smplify + use_contact_in_the_loop: regressor -> SMPL (pose2rot=False) -> rotation matrices to axis-angle ->
    SMPLify-DC (5 + 5 iterations) -> better-fit bookkeeping + dictionary update -> RegressorLoss with the HD contact
    term -> backward into the regressor."""
    from tuch_amd.models.smpl import SMPL
    from tuch_amd.smplify.prior import MaxMixturePrior
    from tuch_amd.smplify.smplifydc import SMPLifyDC
    from tuch_amd.train.loss import RegressorLoss
    from tuch_amd.train.train_module import TUCH
    from tuch_amd.utils.segmentation import BatchBodySegment
    g = _golden()
    batch = int(g['batch'])
    body = make_body(int(g['rings']), int(g['segs']), relax_iters=int(g['relax_iters']))
    train_ds, names = _datasets(g)
    for n in names:
        np.save(tmp_path / (n + '_fits.npy'), g['static_fits_' + n])
    options = _options(g, tmp_path)
    smpl = SMPL(model_data=body, batch_size=batch).to(DEV)
    face_tensor = torch.tensor(body.faces.astype(np.int64), device=DEV)[None].repeat(batch, 1, 1)
    geod = torch.tensor(body.geodesics, device=DEV)
    smplify = SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=int(options.num_smplify_iters), focal_length=5000.,
                        geodistssmpl=geod, geothres=0.3, euclthres=0.02, device=DEV, smpl=smpl,
                        pose_prior=MaxMixturePrior(num_gaussians=8, gmm=body.gmm).to(DEV))
    criterion = RegressorLoss(options=options, device=DEV, num_verts=body.num_verts, faces=face_tensor, geodistssmpl=geod,
                              geothres=0.3, face_tensor=face_tensor,
                              segments=BatchBodySegment(list(body.segments.keys()), face_tensor[0], body.segments),
                              hd_regressor=(body.hd_bary_idx, body.hd_bary_w), hd_faces=body.hd_face_id)
    module = TUCH(options=options, device=DEV, datasets=(train_ds, None), bodymodel=smpl, spin_model=make_regressor(11).to(DEV),
                  regressor=make_regressor(12).to(DEV), optimization=smplify, criterion=criterion, geodistssmpl=geod,
                  contactlists={'classes': [list(p) for p in body.region_pairs], 'csig': dict(body.regions)})
    loss, losses, output = module.forward_train_step(_batch(g))
    loss.backward()
    close_logged(loss.item(), g['loss'], STEP_RTOL, 1e-5, 'train step: loss')
    for k, v in losses.items():
        close_logged(v.cpu().numpy(), g['losses_' + k], STEP_RTOL, 1e-6, 'train step: losses[%s]' % k)
    assert np.array_equal(output['valid_kpts_anno'].cpu().numpy(), g['output_valid_kpts_anno'])
    for k in ('pred_vertices', 'opt_vertices', 'pred_cam_t', 'opt_cam_t', 'spin_vertices', 'spin_cam_t', 'gt_keypoints'):
        want = g['output_' + k]
        close_logged(output[k].cpu().numpy(), want, STEP_RTOL, STEP_RTOL * max(np.abs(want).max(), 1e-3), 'train step: output[%s]' % k)
    close_logged(output['smplifyoptiverts'][-1].cpu().numpy(), g['output_smplifyoptiverts_last'], STEP_RTOL, STEP_RTOL,
                 'train step: optiverts')
    for n in names:
        want = g['fits_after_step_' + n]
        got = module.fits_dict.fits_dict[n].cpu().numpy()
        assert np.array_equal((got != g['static_fits_' + n]).any(1), (want != g['static_fits_' + n]).any(1)), 'updated rows ' + n
        close_logged(got, want, STEP_RTOL, 2 * STEP_RTOL, 'train step: fits table ' + n)
    gw = g['grad_fc_weight']
    close_logged(module.model.fc.weight.grad.cpu().numpy(), gw, STEP_GRAD_RTOL, 0.1 * STEP_GRAD_RTOL * np.abs(gw).max(),
                 'train step: regressor gradient')


def test_forward_train_step_replays_as_a_hip_graph(tmp_path):
    """Without SMPLify in the loop (BASELINE config 4) our part of the training step has no host synchronisation: the
    whole ``forward_train_step`` + backward is captured once and replayed -- same loss and regressor gradient as the
    eager step, and new input values (written in place) are picked up by the replay."""
    from tuch_amd import ops
    # everything, the regressor's parameters included, lives on a created stream: autograd's AccumulateGrad nodes are
    # tied to the stream their parameters were first used on, and capture / replay next to the legacy NULL stream is
    # not reliable on ROCm 7.2
    with ops.off_default_stream(DEV):
        _replay_body(tmp_path)


def _replay_body(tmp_path):
    from tuch_amd.models.smpl import SMPL
    from tuch_amd.smplify.prior import MaxMixturePrior
    from tuch_amd.smplify.smplifydc import SMPLifyDC
    from tuch_amd.train.loss import RegressorLoss
    from tuch_amd.train.train_module import TUCH
    from tuch_amd.utils.segmentation import BatchBodySegment
    g = _golden()
    batch = int(g['batch'])
    body = make_body(int(g['rings']), int(g['segs']), relax_iters=int(g['relax_iters']))
    train_ds, names = _datasets(g)
    for n in names:
        np.save(tmp_path / (n + '_fits.npy'), g['static_fits_' + n])
    options = _options(g, tmp_path)
    options.run_smplify = False
    smpl = SMPL(model_data=body, batch_size=batch).to(DEV)
    face_tensor = torch.tensor(body.faces.astype(np.int64), device=DEV)[None].repeat(batch, 1, 1)
    geod = torch.tensor(body.geodesics, device=DEV)
    smplify = SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=2, focal_length=5000., geodistssmpl=geod, geothres=0.3,
                        euclthres=0.02, device=DEV, smpl=smpl, pose_prior=MaxMixturePrior(num_gaussians=8, gmm=body.gmm).to(DEV))
    criterion = RegressorLoss(options=options, device=DEV, num_verts=body.num_verts, faces=face_tensor, geodistssmpl=geod,
                              geothres=0.3, face_tensor=face_tensor,
                              segments=BatchBodySegment(list(body.segments.keys()), face_tensor[0], body.segments),
                              hd_regressor=(body.hd_bary_idx, body.hd_bary_w), hd_faces=body.hd_face_id)
    module = TUCH(options=options, device=DEV, datasets=(train_ds, None), bodymodel=smpl, spin_model=make_regressor(11).to(DEV),
                  regressor=make_regressor(12).to(DEV), optimization=smplify, criterion=criterion, geodistssmpl=geod,
                  contactlists={'classes': [list(p) for p in body.region_pairs], 'csig': dict(body.regions)})
    inputs = _batch(g)
    params = list(module.model.parameters())
    out = torch.zeros(1, device=DEV)

    def step():
        for q in params:
            q.grad = None
        loss, _, _ = module.forward_train_step(inputs)
        loss.backward()
        out.copy_(loss.detach().reshape(1))

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    eager_loss, eager_grad = out.item(), module.model.fc.weight.grad.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode='thread_local'):
        step()
    captured_grad = module.model.fc.weight.grad           # static tensor of the graph
    graph.replay()
    torch.cuda.synchronize()
    assert_close(out.item(), eager_loss, 1e-5, 1e-7, 'replayed loss')
    assert_close(captured_grad.cpu().numpy(), eager_grad.cpu().numpy(), 1e-4, 1e-6 * float(eager_grad.abs().max()), 'replayed gradient')
    inputs['img'].mul_(0.5)                                # new data, same storage
    graph.replay()
    torch.cuda.synchronize()
    replay_loss = out.item()
    step()
    torch.cuda.synchronize()
    assert replay_loss != eager_loss
    assert_close(replay_loss, out.item(), 1e-5, 1e-7, 'replay with new inputs')


def test_fp32_contact_path_behind_a_bf16_regressor(tmp_path):
    """BASELINE config 5's precision boundary ("bf16 regressor + fp32 contact"): the whole ``forward_train_step`` (SMPLify-DC
    in the loop) under ``torch.autocast('cuda', dtype=torch.bfloat16)`` -- the regressor's linear layer runs in bf16 and hands
    bf16 / bf16-rounded tensors to the body model, whose kernels upcast them (tuch_amd/lbs.py).  Checked: (1) vertices,
    contact term and total are float32 and finite; (2) the contact term equals, to 1e-4, the contact loss of a plain
    float32 run on the SAME upcast regressor outputs; (3) the gradient that reaches the regressor's outputs through the
    body model (cast back to their dtypes by autograd) equals that float32 run's gradient up to bf16 rounding; (4) the
    regressor's (float32) parameters receive finite gradients."""
    from tuch_amd.models.smpl import SMPL
    from tuch_amd.smplify.prior import MaxMixturePrior
    from tuch_amd.smplify.smplifydc import SMPLifyDC
    from tuch_amd.train.loss import RegressorLoss
    from tuch_amd.train.train_module import TUCH
    from tuch_amd.utils.segmentation import BatchBodySegment
    g = _golden()
    batch = int(g['batch'])
    body = make_body(int(g['rings']), int(g['segs']), relax_iters=int(g['relax_iters']))
    train_ds, names = _datasets(g)
    for n in names:
        np.save(tmp_path / (n + '_fits.npy'), g['static_fits_' + n])
    options = _options(g, tmp_path)
    smpl = SMPL(model_data=body, batch_size=batch).to(DEV)
    face_tensor = torch.tensor(body.faces.astype(np.int64), device=DEV)[None].repeat(batch, 1, 1)
    geod = torch.tensor(body.geodesics, device=DEV)
    smplify = SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=int(options.num_smplify_iters), focal_length=5000.,
                        geodistssmpl=geod, geothres=0.3, euclthres=0.02, device=DEV, smpl=smpl,
                        pose_prior=MaxMixturePrior(num_gaussians=8, gmm=body.gmm).to(DEV))
    criterion = RegressorLoss(options=options, device=DEV, num_verts=body.num_verts, faces=face_tensor, geodistssmpl=geod,
                              geothres=0.3, face_tensor=face_tensor,
                              segments=BatchBodySegment(list(body.segments.keys()), face_tensor[0], body.segments),
                              hd_regressor=(body.hd_bary_idx, body.hd_bary_w), hd_faces=body.hd_face_id)
    module = TUCH(options=options, device=DEV, datasets=(train_ds, None), bodymodel=smpl, spin_model=make_regressor(11).to(DEV),
                  regressor=make_regressor(12).to(DEV), optimization=smplify, criterion=criterion, geodistssmpl=geod,
                  contactlists={'classes': [list(p) for p in body.region_pairs], 'csig': dict(body.regions)})
    seen = {}

    def keep(_module, _inputs, out):
        for t in out:
            t.retain_grad()
        seen['out'] = out
    handle = module.model.register_forward_hook(keep)
    # the contact term alone, so that (3) compares exactly the path through the body model and the contact kernels
    valid_seen = {}
    real_contact = criterion.contact_loss

    def spy(verts, valid):
        valid_seen['valid'], valid_seen['verts'] = valid.clone(), verts
        return real_contact(verts, valid)
    criterion.contact_loss = spy
    try:
        with torch.autocast('cuda', dtype=torch.bfloat16):
            loss, losses, output = module.forward_train_step(_batch(g))
        contact = losses['loss_contact']
        assert loss.dtype == torch.float32 and contact.dtype == torch.float32 and output['pred_vertices'].dtype == torch.float32
        assert valid_seen['verts'].dtype == torch.float32
        assert torch.isfinite(loss) and torch.isfinite(contact)
        rot, betas, cam = seen['out']
        assert torch.bfloat16 in (rot.dtype, betas.dtype), 'the regressor did not run under autocast'
        # the contact term's own gradient at the regressor's outputs (bf16 where the outputs are)
        g_rot, g_betas = torch.autograd.grad(contact_of := real_contact(valid_seen['verts'], valid_seen['valid']), [rot, betas],
                                             retain_graph=True)
        assert g_rot.dtype == rot.dtype and g_betas.dtype == betas.dtype
        loss.backward()
        for q in module.model.parameters():
            assert q.grad is not None and q.grad.dtype == torch.float32 and bool(torch.isfinite(q.grad).all())
    finally:
        handle.remove()
        criterion.contact_loss = real_contact
    # the float32 run on the SAME upcast inputs
    rot32 = rot.detach().float().requires_grad_(True)
    betas32 = betas.detach().float().requires_grad_(True)
    out32 = smpl(betas=betas32, body_pose=rot32[:, 1:], global_orient=rot32[:, 0].unsqueeze(1), pose2rot=False)
    assert torch.equal(out32.vertices.detach(), output['pred_vertices'])          # identical upcast inputs -> identical vertices
    contact32 = real_contact(out32.vertices, valid_seen['valid'])
    close_logged(contact.item(), contact32.item(), 1e-4, 1e-9, 'bf16 regressor: contact term vs the float32 run on the same inputs')
    close_logged(contact_of.item(), contact32.item(), 1e-4, 1e-9, 'bf16 regressor: contact term re-evaluated')
    w_rot, w_betas = torch.autograd.grad(contact32, [rot32, betas32])
    for got, want, name in ((g_rot, w_rot, 'rotmat'), (g_betas, w_betas, 'betas')):
        # autograd casts the body model's float32 gradient to the dtype of the tensor it belongs to: one bf16 rounding (2^-8)
        tol = 2.0 ** -7 if got.dtype == torch.bfloat16 else 1e-4
        scale = float(want.abs().max())
        err = float((got.float() - want).abs().max())
        assert err <= tol * scale + 1e-12, (name, err, scale)
