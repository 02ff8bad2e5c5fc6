"""GPU: the HIP SMPL forward/backward (csrc/smpl_lbs.hip, through the C ABI) against the CPU
oracle (float32 restatement of smplx lbs + fp64 truth of the same formulas).
Tolerance: 1e-4 relative on vertices (north_star); observed ~1e-6."""
import numpy as np
import pytest
import torch

from helpers import assert_close
from oracle import lbs as ol
from synthetic import make_body, random_poses

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def bodies():
    return {'tiny': make_body(10, 12, with_geodesics=False), 'full': make_body(84, 82, with_geodesics=False),
            # irregular topology: V = 10 f^2 + 2 is never a multiple of 64 (162, 6762)
            'ico_tiny': make_body(topology='ico', freq=4, with_geodesics=False),
            'ico_full': make_body(topology='ico', freq=26, with_geodesics=False)}


def _smpl(body):
    from tuch_amd.models.smpl import SMPL
    return SMPL(model_data=body).to(DEV)


@pytest.mark.parametrize('size,batch', [('tiny', 1), ('tiny', 3), ('tiny', 17), ('full', 2), ('full', 64), ('ico_tiny', 5),
                                        ('ico_full', 3), ('ico_full', 64)])
def test_forward_matches_oracle(bodies, size, batch):
    body = bodies[size]
    bp, go, be = random_poses(batch, 21 + batch)
    t = lambda a: torch.tensor(a, device=DEV)
    out = _smpl(body)(betas=t(be), body_pose=t(bp), global_orient=t(go))
    t64 = lambda a: torch.tensor(a, dtype=torch.float64)
    v64, j64 = ol.smpl_forward(ol.model_tensors(body, torch.float64), t64(be), t64(bp), t64(go))
    assert out.vertices.shape == (batch, body.num_verts, 3) and out.joints.shape == (batch, 49, 3)
    assert_close(out.vertices.cpu().numpy(), v64.numpy(), 1e-4, 5e-6, 'verts vs fp64')
    assert_close(out.joints.cpu().numpy(), j64.numpy(), 1e-4, 5e-6, 'joints vs fp64')
    v32, j32 = ol.smpl_forward(ol.model_tensors(body), torch.tensor(be), torch.tensor(bp), torch.tensor(go))
    assert_close(out.vertices.cpu().numpy(), v32.numpy(), 1e-5, 5e-6, 'verts vs f32 oracle')


@pytest.mark.parametrize('size,batch', [('tiny', 3), ('full', 5), ('ico_tiny', 2), ('ico_full', 5)])
@pytest.mark.parametrize('pose2rot', [True, False])
def test_backward_matches_oracle_autograd(bodies, size, batch, pose2rot):
    body = bodies[size]
    bp, go, be = random_poses(batch, 33)
    rng = np.random.default_rng(5)
    gv = rng.standard_normal((batch, body.num_verts, 3)).astype(np.float32)
    gj = rng.standard_normal((batch, 49, 3)).astype(np.float32)
    m64 = ol.model_tensors(body, torch.float64)
    t64 = lambda a: torch.tensor(a, dtype=torch.float64)
    be64 = t64(be).requires_grad_(True)
    full64 = torch.cat([t64(go), t64(bp)], 1)
    if pose2rot:
        pose64 = full64.clone().requires_grad_(True)
    else:
        pose64 = ol.rodrigues(full64.reshape(-1, 3)).reshape(batch, 24, 3, 3).clone().requires_grad_(True)
    v, j = ol.lbs(be64, pose64, m64, pose2rot=pose2rot)
    picked = v[:, m64['extra_vertex_ids']]
    extra = torch.einsum('bvk,jv->bjk', v, m64['J_regressor_extra'])
    joints = torch.cat([j, picked, extra], 1)[:, m64['joint_map']]
    ((v * t64(gv)).sum() + (joints * t64(gj)).sum()).backward()

    smpl = _smpl(body)
    t = lambda a: torch.tensor(a, device=DEV)
    be_d = t(be).requires_grad_(True)
    if pose2rot:
        go_d, bp_d = t(go).requires_grad_(True), t(bp).requires_grad_(True)
        out = smpl(betas=be_d, body_pose=bp_d, global_orient=go_d)
    else:
        rot = pose64.detach().to(torch.float32).to(DEV)
        go_d, bp_d = rot[:, :1].clone().requires_grad_(True), rot[:, 1:].clone().requires_grad_(True)
        out = smpl(betas=be_d, body_pose=bp_d, global_orient=go_d, pose2rot=False)
    ((out.vertices * t(gv)).sum() + (out.joints * t(gj)).sum()).backward()
    got_pose = torch.cat([go_d.grad.reshape(batch, -1), bp_d.grad.reshape(batch, -1)], 1).cpu().numpy()
    want_pose = pose64.grad.reshape(batch, -1).numpy()
    from helpers import grad_close
    grad_close(got_pose, want_pose, 2e-5, 'lbs %s B=%d pose2rot=%s grad pose' % (size, batch, pose2rot))
    grad_close(be_d.grad.cpu().numpy(), be64.grad.numpy(), 2e-5, 'lbs %s B=%d pose2rot=%s grad betas' % (size, batch, pose2rot))


def test_joints_only_and_verts_only_gradients(bodies):
    body = bodies['tiny']
    smpl = _smpl(body)
    bp, go, be = [torch.tensor(a, device=DEV) for a in random_poses(2, 3)]
    for which in ('joints', 'vertices'):
        bp_ = bp.clone().requires_grad_(True)
        out = smpl(betas=be, body_pose=bp_, global_orient=go)
        getattr(out, which).sum().backward()
        assert torch.isfinite(bp_.grad).all() and bp_.grad.abs().sum() > 0


@pytest.mark.parametrize('pose2rot', [True, False])
def test_concatenated_pose_and_the_two_pose_tensors_agree(bodies, pose2rot):
    """tuch/models/smpl.py:44-47 concatenates global_orient and body_pose; the kernels read the two tensors directly.
    Row views of ONE [B,72] / [B,24,9] pose (strided rows) must give the same bits, and the same gradient."""
    from tuch_amd import lbs
    body = bodies['tiny']
    smpl = _smpl(body)
    bp, go, be = [torch.tensor(a, device=DEV) for a in random_poses(5, 11)]
    if not pose2rot:
        rot = ol.rodrigues(torch.cat([go, bp], 1).reshape(-1, 3).cpu()).reshape(5, 24, 9).to(DEV)
        go, bp = rot[:, :1].reshape(5, 9).contiguous(), rot[:, 1:].reshape(5, 207).contiguous()
    full = torch.cat([go, bp], 1).requires_grad_(True)
    go_, bp_ = go.clone().requires_grad_(True), bp.clone().requires_grad_(True)
    v1, j1 = lbs.smpl_forward(smpl, be, full, pose2rot)
    out = smpl(betas=be, body_pose=bp_, global_orient=go_, pose2rot=pose2rot, return_full_pose=True)
    assert torch.equal(v1, out.vertices) and torch.equal(j1, out.joints)
    assert torch.equal(out.full_pose.reshape(5, -1), full.detach())
    w = torch.linspace(0.5, 1.5, v1.numel(), device=DEV).reshape(v1.shape)
    ((v1 * w).sum() + j1.sum()).backward()
    ((out.vertices * w).sum() + out.joints.sum()).backward()
    assert torch.equal(full.grad, torch.cat([go_.grad, bp_.grad], 1))


@pytest.mark.parametrize('size,batch', [('tiny', 5), ('full', 33), ('ico_full', 64), ('full', 16)])
def test_sparse_skinning_is_the_dense_sum_bit_for_bit_and_every_batch_form_agrees(bodies, size, batch, monkeypatch):
    """(1) A model whose vertices have at most four non-zero skinning weights (SMPL's own) takes the sparse skinning kernels:
    the same fma chain as the dense 24-joint loop minus its fma(0, A, T) = T terms -- vertices, joints and gradients are
    the dense kernels' BITS (TUCH_SKIN_DENSE=1 at model creation keeps the dense form).  (2) A body with more than four
    weights per vertex falls back to the dense kernels and still matches the oracle.  The batch sizes cover the three
    forms of the blend kernel (one / two body tiles per wavefront, K split over the workgroup)."""
    body = bodies[size]
    assert int((body.lbs_weights != 0).sum(1).max()) <= 4
    bp, go, be = random_poses(batch, 77)
    t = lambda a: torch.tensor(a, device=DEV)
    rng = np.random.default_rng(9)
    gv, gj = t(rng.standard_normal((batch, body.num_verts, 3)).astype(np.float32)), t(rng.standard_normal((batch, 49, 3)).astype(np.float32))

    def run(dense):
        if dense:
            monkeypatch.setenv('TUCH_SKIN_DENSE', '1')
        else:
            monkeypatch.delenv('TUCH_SKIN_DENSE', raising=False)
        smpl = _smpl(body)
        b_, p_, g_ = t(be).requires_grad_(True), t(bp).requires_grad_(True), t(go).requires_grad_(True)
        out = smpl(betas=b_, body_pose=p_, global_orient=g_)
        torch.autograd.backward([out.vertices, out.joints], [gv, gj])
        return [x.detach().clone() for x in (out.vertices, out.joints, b_.grad, p_.grad, g_.grad)]
    sparse, dense = run(False), run(True)
    for a, b, name in zip(sparse, dense, ('verts', 'joints', 'g_betas', 'g_body_pose', 'g_global_orient')):
        assert torch.equal(a, b), name
    # more than four weights per vertex: the dense kernels, against the oracle
    import copy
    fat = copy.copy(body)
    w = body.lbs_weights.astype(np.float64) + 0.02
    fat.lbs_weights = (w / w.sum(1, keepdims=True)).astype(np.float32)
    monkeypatch.delenv('TUCH_SKIN_DENSE', raising=False)
    n = min(batch, 3)
    out = _smpl(fat)(betas=t(be[:n]), body_pose=t(bp[:n]), global_orient=t(go[:n]))
    t64 = lambda a: torch.tensor(a, dtype=torch.float64)
    v64, _ = ol.smpl_forward(ol.model_tensors(fat, torch.float64), t64(be[:n]), t64(bp[:n]), t64(go[:n]))
    assert_close(out.vertices.cpu().numpy(), v64.numpy(), 1e-4, 5e-6, 'dense-weight body vs fp64')


def test_leaf_views_of_a_plain_base_and_watched_views_keep_their_own_gradients(bodies):
    """A fitting setup: pose = zeros(B,72) WITHOUT grad, the two parameters are its row views made leaves with
    requires_grad_().  They look like the shared rows of one base but autograd owes each view its own .grad; likewise a
    view somebody put a hook / retain_grad() on.  Same numbers as two independent tensors."""
    body = bodies['tiny']
    smpl = _smpl(body)
    bp, go, be = [torch.tensor(a, device=DEV) for a in random_poses(5, 31)]
    w = torch.linspace(0.5, 1.5, 5 * body.num_verts * 3, device=DEV).reshape(5, body.num_verts, 3)

    def objective(g, p):
        out = smpl(betas=be, body_pose=p, global_orient=g)
        return (out.vertices * w).sum() + out.joints.sum()
    g0, p0 = go.clone().requires_grad_(True), bp.clone().requires_grad_(True)
    objective(g0, p0).backward()
    # leaf views of a base without grad
    pose = torch.cat([go, bp], 1).contiguous()
    gl, pl = pose[:, :3].requires_grad_(), pose[:, 3:].requires_grad_()
    assert gl.is_leaf and pl.is_leaf and not pose.requires_grad
    objective(gl, pl).backward()
    assert torch.equal(gl.grad, g0.grad) and torch.equal(pl.grad, p0.grad)
    # non-leaf views of a differentiable base with a hook and a retained gradient on them
    base = torch.cat([go, bp], 1).contiguous().requires_grad_(True)
    gv, pv = base[:, :3], base[:, 3:]
    seen = []
    gv.register_hook(lambda grad: seen.append(grad.clone()))
    pv.retain_grad()
    objective(gv, pv).backward()
    assert len(seen) == 1 and torch.equal(seen[0], g0.grad) and torch.equal(pv.grad, p0.grad)
    assert torch.equal(base.grad[:, :3], g0.grad) and torch.equal(base.grad[:, 3:], p0.grad)
    # unwatched views of a differentiable base: the one-gradient path, same bits
    base2 = torch.cat([go, bp], 1).contiguous().requires_grad_(True)
    objective(base2[:, :3], base2[:, 3:]).backward()
    assert torch.equal(base2.grad, base.grad)


def test_train_style_slices_of_one_rotation_tensor_take_one_gradient(bodies):
    """train_module.py:202-204 passes pred_rotmat[:, 1:] and pred_rotmat[:, 0].unsqueeze(1): two views of ONE [B,24,3,3]
    tensor.  The module hands the kernels that tensor's rows and returns ONE gradient for it (autograd would zero-fill and
    copy two slice gradients and add them): same bits as two separate tensors, and anything that is not exactly
    (orientation first, pose behind it, nothing else) in one contiguous base keeps the two-tensor path."""
    from tuch_amd import lbs
    body = bodies['tiny']
    smpl = _smpl(body)
    bp, go, be = [torch.tensor(a, device=DEV) for a in random_poses(6, 23)]
    rot = ol.rodrigues(torch.cat([go, bp], 1).reshape(-1, 3).cpu()).reshape(6, 24, 3, 3).to(DEV)
    w = torch.linspace(0.5, 1.5, 6 * body.num_verts * 3, device=DEV).reshape(6, body.num_verts, 3)

    def run(make):
        leaf = rot.clone().requires_grad_(True)
        g, p = make(leaf)
        out = smpl(betas=be, body_pose=p, global_orient=g, pose2rot=False)
        ((out.vertices * w).sum() + out.joints.sum()).backward()
        return out.vertices.detach(), out.joints.detach(), leaf.grad
    ref = run(lambda t: (t[:, :1].clone(), t[:, 1:].clone()))                       # two tensors of their own
    shared = lambda t: lbs._shared_rows(*t, 9) is not None
    leaf = rot.clone().requires_grad_(True)
    forms = {'select + unsqueeze (the reference)': lambda t: (t[:, 0].unsqueeze(1), t[:, 1:]),
             'two slices': lambda t: (t[:, :1], t[:, 1:]),
             'rows of the flat tensor': lambda t: (t.view(6, 216)[:, :9], t.view(6, 216)[:, 9:])}
    for name, make in forms.items():
        assert shared(make(leaf)), name
        got = run(make)
        for a, b in zip(got, ref):
            assert torch.equal(a, b), name
    # not the whole base / the wrong order / different bases / a copy: the two-tensor path, same numbers
    wide = torch.cat([rot.reshape(6, 216), torch.zeros(6, 9, device=DEV)], 1).requires_grad_(True)
    assert not shared((wide[:, :9], wide[:, 9:216]))
    assert not shared((leaf[:, 1:2], leaf[:, 1:]))
    assert not shared((rot.clone()[:, :1], leaf[:, 1:]))
    assert not shared((leaf[:, :1].clone(), leaf[:, 1:]))
    other = rot.clone().requires_grad_(True)
    out = smpl(betas=be, body_pose=other[:, 1:], global_orient=leaf[:, :1], pose2rot=False)
    ((out.vertices * w).sum() + out.joints.sum()).backward()
    assert torch.equal(out.vertices.detach(), ref[0])
    assert torch.equal(leaf.grad[:, :1], ref[2][:, :1]) and torch.equal(other.grad[:, 1:], ref[2][:, 1:])
    assert float(leaf.grad[:, 1:].abs().max()) == 0.0 and float(other.grad[:, :1].abs().max()) == 0.0
    # no gradient wanted: nothing changes
    with torch.no_grad():
        o2 = smpl(betas=be, body_pose=rot[:, 1:], global_orient=rot[:, :1], pose2rot=False)
    assert torch.equal(o2.vertices, ref[0])
