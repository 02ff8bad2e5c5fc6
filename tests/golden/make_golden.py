#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE ITSELF
(muelea/tuch at /root/reference, imported, never copied) on synthetic inputs.

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is shimmed so the reference's hot path runs on a CPU-only box:
  * torch.cuda.LongTensor = torch.LongTensor (contact.py:30-31 hard-codes use_cuda=True)
  * contact_fitting_loss is called with device='cpu' (losses.py:43 defaults to 'cuda')
  * stub modules for the un-shipped ``trimesh`` and ``data.essentials.segments.smpl.segm_utils``
    whose *contents* are our synthetic segments; BodySegment / BatchBodySegment /
    RegressorLoss / MaxMixturePrior are then constructed by their real __init__
    from temp files written in the formats they load (config.HD_MODEL_DIR, PRIOR_FOLDER).
tuch.models.smpl (needs smplx) and tuch.train.train_module (needs smplx, torchgeometry,
constants) cannot be imported; contact_from_verts is reproduced by calling the reference's
batch_pairwise_dist from the same loop as train_module.py:83-90.

Every fixture stores its inputs next to the expected outputs, so the tests never
depend on regenerating floating-point inputs bit-for-bit on another host.
"""
import os
import pickle
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(1, REF)

import numpy as np
import torch

import golden_io as gio
from oracle import lbs as olbs
from synthetic import dense_hd_regressor, make_body, random_poses, through_pose

torch.cuda.LongTensor = torch.LongTensor  # F7 shim
torch.manual_seed(0)

# ------------------------------------------------------------------ stub modules
_STATE = {'segments': {}, 'num_verts': 0}


def _install_stubs():
    tm = types.ModuleType('trimesh')

    def load(path, process=False):
        name = os.path.basename(path)[len('smpl_segment_'):-len('.ply')]
        colors = np.zeros((_STATE['num_verts'], 4), np.uint8)
        colors[_STATE['segments'][name]['vidx'], 0] = 255
        mesh = types.SimpleNamespace(visual=types.SimpleNamespace(vertex_colors=colors))
        return mesh
    tm.load = load
    sys.modules['trimesh'] = tm
    chain = ['data', 'data.essentials', 'data.essentials.segments', 'data.essentials.segments.smpl']
    for name in chain:
        mod = types.ModuleType(name)
        mod.__path__ = []
        sys.modules[name] = mod
    su = types.ModuleType('data.essentials.segments.smpl.segm_utils')
    su.segments = {}
    sys.modules[su.__name__] = su
    sys.modules['data.essentials.segments.smpl'].segm_utils = su
    return su


_SEGM_UTILS = _install_stubs()

from configs import config as ref_config                      # noqa: E402
from tuch.smplify import losses as ref_losses                 # noqa: E402
from tuch.smplify.prior import MaxMixturePrior                # noqa: E402
from tuch.train import loss as ref_train_loss                 # noqa: E402
from tuch.utils import contact as ref_contact                 # noqa: E402
from tuch.utils import geometry as ref_geometry               # noqa: E402
from tuch.utils import segmentation as ref_segmentation       # noqa: E402


def _use_body(body):
    _STATE['segments'] = body.segments
    _STATE['num_verts'] = body.num_verts
    _SEGM_UTILS.segments.clear()
    for name, seg in body.segments.items():
        _SEGM_UTILS.segments[name] = {k: [int(x) for x in v] for k, v in seg['bands'].items()}


def _posed_verts(body, batch, seed):
    m = olbs.model_tensors(body)
    bp, go, be = random_poses(batch, seed)
    v, j = olbs.smpl_forward(m, torch.tensor(be), torch.tensor(bp), torch.tensor(go))
    return v.numpy().astype(np.float32), j.numpy().astype(np.float32), bp, go, be


def _prior(tmp, gmm):
    with open(os.path.join(tmp, 'gmm_08.pkl'), 'wb') as f:
        pickle.dump({k: np.asarray(v, np.float64) for k, v in gmm.items()}, f)
    return MaxMixturePrior(prior_folder=tmp, num_gaussians=8, dtype=torch.float32)


def _common_inputs(body, out):
    out['faces'] = body.faces
    gio.pack_mask(body.geodesics > ref_config.geothres, out)
    gio.pack_segments(body.segments, out)
    gio.pack_regions(body.regions, body.region_pairs, out)
    out['hd_idx'] = body.hd_bary_idx
    out['hd_w'] = body.hd_bary_w
    out['hd_face'] = body.hd_face_id


def contact_case(tag, body_kw, batch, seed, store_dense):
    """K1-K6 + a6/a7/a8 of SURVEY.md §8a on one synthetic model."""
    body = make_body(**body_kw)
    _use_body(body)
    verts_np, joints_np, bp, go, be = _posed_verts(body, batch, seed)
    out = {}
    _common_inputs(body, out)
    out['verts'] = verts_np
    out['euclthres'] = np.float32(ref_config.euclthres)
    num_verts = body.num_verts
    face_tensor = torch.tensor(body.faces, dtype=torch.long)[None].repeat(batch, 1, 1)
    geomask = torch.tensor(body.geodesics) > ref_config.geothres
    names = list(body.segments.keys())
    segments = ref_segmentation.BatchBodySegment(names, face_tensor[0])

    verts = torch.tensor(verts_np)
    # --- a1/a2: pairwise + masked argmin (losses.py:76-78,92-93)
    mins, args = [], []
    for b in range(batch):
        P = ref_contact.batch_pairwise_dist(verts[[b]], verts[[b]], squared=True)
        if store_dense and b == 0:
            out['pairwise_b0'] = P[0].numpy().copy()
        P[:, ~geomask] = float('inf')
        mn, arg = torch.min(P, dim=1)
        mins.append(mn[0].numpy())
        args.append(arg[0].numpy())
    out['v2v_min'] = np.stack(mins)
    out['v2v_argmin'] = np.stack(args)
    # --- a3/a4: solid angles + winding (contact.py:49-147)
    ws = []
    for b in range(batch):
        tris = verts[b][face_tensor[0]]
        if store_dense and b == 0:
            out['solid_angles_b0'] = ref_contact.solid_angles(verts[[b]], tris[None])[0].numpy()
        ws.append(ref_contact.winding_numbers(verts[[b]], tris[None])[0].numpy())
    out['winding'] = np.stack(ws)
    # --- a5: segments (segmentation.py:117-124)
    seg_ext = []
    for b in range(batch):
        exts = segments.batch_has_self_isec(verts[[b]])
        seg_ext.append(np.concatenate([e.numpy().astype(np.uint8) for e in exts]))
    out['segment_exterior'] = np.stack(seg_ext)
    out['segment_faces_count'] = np.asarray(
        [segments.segmentation[n].segment_faces.shape[0] for n in names], np.int64)
    gio.pack_ragged('segment_faces', [segments.segmentation[n].segment_faces.numpy().ravel()
                                      for n in names], out)

    # --- a6: contact_fitting_loss (losses.py:34-123)
    rng = np.random.Generator(np.random.PCG64(seed + 7))
    num_pairs = len(body.region_pairs)
    gt = (rng.random((batch, num_pairs)) < 0.05).astype(np.float32)
    gt[:, 0] = 1.0
    has_dc = np.ones(batch, bool)
    has_dc[-1] = False
    ignore = np.zeros(batch, bool)
    cam_t = np.tile(np.array([[0.0, 0.0, 20.0]], np.float32), (batch, 1)) \
        + 0.1 * rng.standard_normal((batch, 3)).astype(np.float32)
    cam_c = np.zeros((batch, 2), np.float32)
    j2d = ref_geometry.perspective_projection(
        torch.tensor(joints_np), torch.eye(3)[None].expand(batch, -1, -1), torch.tensor(cam_t),
        5000., torch.tensor(cam_c)).numpy() + 3.0 * rng.standard_normal((batch, 49, 2)).astype(np.float32)
    conf = (0.5 + 0.5 * rng.random((batch, 49))).astype(np.float32)
    out.update(gt_contact=gt, has_discrete_contact=has_dc, ignore_idxs=ignore, camera_t=cam_t,
               camera_center=cam_c, joints_2d=j2d.astype(np.float32), joints_conf=conf,
               model_joints=joints_np, body_pose=bp, global_orient=go, betas=be,
               contact_loss_weight=np.float32(2000.0))
    for k, v in body.gmm.items():
        out['gmm_' + k] = np.asarray(v, np.float64)
    cdict = {'classes': [list(p) for p in body.region_pairs],
             'csig': {k: v for k, v in body.regions.items()}}
    with tempfile.TemporaryDirectory() as tmp:
        prior = _prior(tmp, body.gmm)
        zero_prior = lambda pose, betas: torch.zeros(pose.shape[0])
        for eu_tag, eucl in (('e0', 0.0), ('e2', ref_config.euclthres)):
            for sg_tag, sg in (('nos', None), ('seg', segments)):
                for full in (False, True):
                    v = torch.tensor(verts_np, requires_grad=True)
                    mj = torch.tensor(joints_np, requires_grad=True)
                    pose = torch.tensor(bp, requires_grad=True)
                    c = torch.tensor(conf if full else np.zeros_like(conf))
                    loss = ref_losses.contact_fitting_loss(
                        pose, torch.tensor(go), None, None, torch.tensor(be), mj, geomask, eucl,
                        torch.tensor(cam_t), torch.tensor(cam_c), torch.tensor(j2d), c,
                        prior if full else zero_prior, cdict, [torch.tensor(gt), None],
                        torch.tensor(ignore), torch.tensor(has_dc), v, face_tensor=face_tensor,
                        device='cpu', focal_length=5000., contact_loss_weight=2000.0, segments=sg)
                    loss.backward()
                    key = 'smplify_%s_%s_%s' % (eu_tag, sg_tag, 'full' if full else 'contact')
                    out[key + '_loss'] = np.float64(loss.item())
                    out[key + '_grad_verts'] = v.grad.numpy()
                    if full:
                        out[key + '_grad_joints'] = mj.grad.numpy()
                        out[key + '_grad_pose'] = pose.grad.numpy()

        # --- a11 pieces (geometry.py:83-111, losses.py:25-32,125-152,164-198, prior.py:117-132)
        out['prior_values'] = prior(torch.tensor(bp), torch.tensor(be)).numpy()
        out['projected_joints'] = ref_geometry.perspective_projection(
            torch.tensor(joints_np), torch.eye(3)[None].expand(batch, -1, -1), torch.tensor(cam_t),
            5000., torch.tensor(cam_c)).numpy()
        out['gmof_values'] = ref_losses.gmof(torch.tensor(j2d) - torch.tensor(out['projected_joints']),
                                             100.).numpy()
        smpl_out = types.SimpleNamespace(joints=torch.tensor(joints_np), betas=torch.tensor(be))
        cam_est = torch.tensor(cam_t) + 0.05
        out['camera_t_est'] = cam_est.numpy()
        out['camera_fitting_loss'] = np.float64(ref_losses.camera_fitting_loss(
            smpl_out, torch.tensor(cam_t), cam_est, torch.tensor(cam_c), torch.tensor(j2d),
            torch.tensor(conf), focal_length=5000., shape_prior_weight=1.0).item())
        out['body_fitting_reprojection'] = ref_losses.body_fitting_loss(
            torch.tensor(bp), torch.tensor(be), torch.tensor(joints_np), torch.tensor(cam_t),
            torch.tensor(cam_c), torch.tensor(j2d), torch.tensor(conf), prior, focal_length=5000.,
            output='reprojection').numpy()
        out['body_fitting_sum'] = np.float64(ref_losses.body_fitting_loss(
            torch.tensor(bp), torch.tensor(be), torch.tensor(joints_np), torch.tensor(cam_t),
            torch.tensor(cam_c), torch.tensor(j2d), torch.tensor(conf), prior, focal_length=5000.).item())

        # --- a7: RegressorLoss.contact_loss (loss.py:240-317), both branches
        ref_config.HD_MODEL_DIR = tmp
        np.save(os.path.join(tmp, 'smpl_neutral_hd_vert_regressor.npy'), dense_hd_regressor(body))
        with open(os.path.join(tmp, 'smpl_neutral_hd_sample_from_mesh_out.pkl'), 'wb') as f:
            pickle.dump({'faces_vert_is_sampled_from': body.hd_face_id}, f)
        valid = np.ones(batch, bool)
        if batch > 2:
            valid[1] = False
        out['valid_fit'] = valid
        for use_hd in (False, True):
            crit = ref_train_loss.RegressorLoss(
                options=types.SimpleNamespace(contact_loss_weight=1.0), device='cpu',
                num_verts=num_verts, faces=face_tensor, geodistssmpl=torch.tensor(body.geodesics),
                geothres=ref_config.geothres, euclthres=ref_config.euclthres,
                face_tensor=face_tensor, use_hd=use_hd)
            v = torch.tensor(verts_np, requires_grad=True)
            loss = crit.contact_loss(v, torch.tensor(valid))
            loss.backward()
            key = 'train_hd' if use_hd else 'train_plain'
            out[key + '_loss'] = np.float64(loss.item())
            out[key + '_grad_verts'] = v.grad.numpy()

    # --- f1: EFT variant, EFTLoss.contact_loss (tuch/eft/loss.py:129-181); the reference tests the
    # segments with the whole batch (:150), which only works for batch 1 -> one call per body
    from tuch.eft import loss as ref_eft
    eft = ref_eft.EFTLoss.__new__(ref_eft.EFTLoss)
    torch.nn.Module.__init__(eft)
    eft.device, eft.options = 'cpu', types.SimpleNamespace(batch_size=1)
    eft.face_tensor, eft.geomask, eft.cdict, eft.segments = face_tensor[:1], geomask, cdict, segments
    eft_loss, eft_grad = [], []
    for b in range(batch):
        v = torch.tensor(verts_np[b:b + 1], requires_grad=True)
        l = eft.contact_loss(torch.tensor(gt[b:b + 1]), v)
        l.backward()
        eft_loss.append(l.item())
        eft_grad.append(v.grad.numpy()[0])
    out['eft_loss'] = np.asarray(eft_loss, np.float64)
    out['eft_grad_verts'] = np.stack(eft_grad)

    # --- a8: contact_from_verts (train_module.py:83-90 loop around the reference's own op)
    pc = np.zeros((batch, num_pairs), np.float32)
    for k, (ra, rb) in enumerate(body.region_pairs):
        d = ref_contact.batch_pairwise_dist(verts[:, body.regions[ra], :], verts[:, body.regions[rb], :],
                                            squared=True)
        pc[:, k] = torch.min(d.view(batch, -1), dim=1)[0].numpy()
    out['contact_from_verts'] = pc
    path = os.path.join(HERE, 'contact_%s.npz' % tag)
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024))


def _fullsize_poses(tag, body):
    """(verts [B,V,3], ignore_idxs [B]) of a full-size fixture."""
    if tag == 'full2':
        # body 0: the left forearm pushed THROUGH the trunk / hip (157 of its 306 vertices inside, 230 interior
        # vertices in all), body 1: an ordinary pose that the caller marks ignore_idxs (losses.py:74)
        m = olbs.model_tensors(body)
        bp, go, be = through_pose(2, 2003)
        v, _ = olbs.smpl_forward(m, torch.tensor(be), torch.tensor(bp), torch.tensor(go))
        return v.numpy().astype(np.float32), np.array([False, True])
    verts_np, _, _, _, _ = _posed_verts(body, 1, 2002)
    return verts_np, np.array([False])


def fullsize_case(tag='full', body_kw=None):
    """SMPL-sized bodies (V=6890, F=13776; 'ico_full': the irregular mesh, V=6762): winding numbers, masked v2v and
    the SMPLify contact term with segments, straight from the reference (peak RSS ~8 GB per body)."""
    body = make_body(**(body_kw or FULL_BODIES[tag]))
    _use_body(body)
    verts_np, ignore = _fullsize_poses(tag, body)
    batch = verts_np.shape[0]
    out = {}
    _common_inputs(body, out)
    out['verts'] = verts_np
    out['ignore_idxs'] = ignore
    face_tensor = torch.tensor(body.faces, dtype=torch.long)[None]
    geomask = torch.tensor(body.geodesics) > ref_config.geothres
    verts = torch.tensor(verts_np)
    names = list(body.segments.keys())
    segments = ref_segmentation.BatchBodySegment(names, face_tensor[0])
    ws, mins, args, seg_ext = [], [], [], []
    for b in range(batch):
        tris = verts[b][face_tensor[0]]
        ws.append(ref_contact.winding_numbers(verts[[b]], tris[None])[0].numpy())
        P = ref_contact.batch_pairwise_dist(verts[[b]], verts[[b]], squared=True)
        P[:, ~geomask] = float('inf')
        mn, arg = torch.min(P, dim=1)
        mins.append(mn[0].numpy())
        args.append(arg[0].numpy())
        del P
        exts = segments.batch_has_self_isec(verts[[b]])
        seg_ext.append(np.concatenate([e.numpy().astype(np.uint8) for e in exts]))
    out['winding'] = np.stack(ws)
    out['v2v_min'] = np.stack(mins)
    out['v2v_argmin'] = np.stack(args)
    out['segment_exterior'] = np.stack(seg_ext)
    zero_prior = lambda pose, betas: torch.zeros(pose.shape[0])
    cdict = {'classes': [list(p) for p in body.region_pairs], 'csig': dict(body.regions)}
    gt = np.zeros((batch, len(body.region_pairs)), np.float32)
    gt[:, :3] = 1.0
    out['gt_contact'] = gt
    for eu_tag, eucl in (('e0', 0.0), ('e2', ref_config.euclthres)):
        v = torch.tensor(verts_np, requires_grad=True)
        loss = ref_losses.contact_fitting_loss(
            torch.zeros(batch, 69), torch.zeros(batch, 3), None, None, torch.zeros(batch, 10),
            torch.zeros(batch, 49, 3) + 1.0, geomask, eucl, torch.tensor([[0., 0., 20.]]).repeat(batch, 1),
            torch.zeros(batch, 2), torch.zeros(batch, 49, 2), torch.zeros(batch, 49), zero_prior, cdict,
            [torch.tensor(gt), None], torch.tensor(ignore), torch.ones(batch, dtype=torch.bool), v,
            face_tensor=face_tensor, device='cpu', contact_loss_weight=2000.0, segments=segments)
        loss.backward()
        out['smplify_%s_seg_contact_loss' % eu_tag] = np.float64(loss.item())
        out['smplify_%s_seg_contact_grad_verts' % eu_tag] = v.grad.numpy()
        print(tag, eu_tag, loss.item(), flush=True)
    out['contact_loss_weight'] = np.float32(2000.0)
    out['euclthres'] = np.float32(ref_config.euclthres)
    path = os.path.join(HERE, 'contact_%s.npz' % tag)
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024))


def fullsize_train_case(tag='full'):
    """The SMPL-sized bodies of ``fullsize_case(tag)`` (same vertices) through the reference's
    RegressorLoss.contact_loss, plain and HD branch with all N_hd = 3 F = 41 328 HD points
    (tuch/train/loss.py:240-317), and through EFTLoss.contact_loss (tuch/eft/loss.py:129-181).
    Inputs live in contact_full.npz; this file holds the expected outputs only."""
    body = make_body(**FULL_BODIES[tag])
    _use_body(body)
    verts_np, _ = _fullsize_poses(tag, body)
    assert np.array_equal(verts_np, gio.load('contact_%s.npz' % tag)['verts'])
    batch = verts_np.shape[0]
    out = {}
    face_tensor = torch.tensor(body.faces, dtype=torch.long)[None]
    geomask = torch.tensor(body.geodesics) > ref_config.geothres
    valid = torch.ones(batch, dtype=torch.bool)
    with tempfile.TemporaryDirectory() as tmp:
        ref_config.HD_MODEL_DIR = tmp
        np.save(os.path.join(tmp, 'smpl_neutral_hd_vert_regressor.npy'), dense_hd_regressor(body))
        with open(os.path.join(tmp, 'smpl_neutral_hd_sample_from_mesh_out.pkl'), 'wb') as f:
            pickle.dump({'faces_vert_is_sampled_from': body.hd_face_id}, f)
        for use_hd in (False, True):
            crit = ref_train_loss.RegressorLoss(
                options=types.SimpleNamespace(contact_loss_weight=1.0), device='cpu',
                num_verts=body.num_verts, faces=face_tensor, geodistssmpl=torch.tensor(body.geodesics),
                geothres=ref_config.geothres, euclthres=ref_config.euclthres,
                face_tensor=face_tensor, use_hd=use_hd)
            v = torch.tensor(verts_np, requires_grad=True)
            loss = crit.contact_loss(v, valid)
            loss.backward()
            key = 'train_hd' if use_hd else 'train_plain'
            out[key + '_loss'] = np.float64(loss.item())
            out[key + '_grad_verts'] = v.grad.numpy()
            print(key, loss.item(), flush=True)
            del crit
    from tuch.eft import loss as ref_eft
    names = list(body.segments.keys())
    segments = ref_segmentation.BatchBodySegment(names, face_tensor[0])
    cdict = {'classes': [list(p) for p in body.region_pairs], 'csig': dict(body.regions)}
    gt = gio.load('contact_%s.npz' % tag)['gt_contact']
    eft = ref_eft.EFTLoss.__new__(ref_eft.EFTLoss)
    torch.nn.Module.__init__(eft)
    eft.device, eft.options = 'cpu', types.SimpleNamespace(batch_size=1)
    eft.face_tensor, eft.geomask, eft.cdict, eft.segments = face_tensor, geomask, cdict, segments
    eft_loss, eft_grad = [], []
    for b in range(batch):              # the reference tests the segments with the whole batch (:150): one call per body
        v = torch.tensor(verts_np[b:b + 1], requires_grad=True)
        l = eft.contact_loss(torch.tensor(gt[b:b + 1]), v)
        l.backward()
        eft_loss.append(l.item())
        eft_grad.append(v.grad.numpy()[0])
        print('eft', l.item(), flush=True)
    out['eft_loss'] = np.asarray(eft_loss, np.float64)
    out['eft_grad_verts'] = np.stack(eft_grad)
    path = os.path.join(HERE, 'contact_%s_train.npz' % tag)
    np.savez_compressed(path, **out)
    print('wrote', path, '%.1f KB' % (os.path.getsize(path) / 1024))


BODIES = {
    'small': (dict(rings=10, segs=12), 2, 1001, True),
    'medium': (dict(rings=40, segs=40), 3, 1002, False),
    # irregular topology: geodesic icosahedron + random edge flips (valence 4-9, V not a multiple of 64), painted
    # segments with ragged boundaries and stray vertices
    'ico_small': (dict(topology='ico', freq=4), 2, 1003, True),
    'ico_medium': (dict(topology='ico', freq=13), 3, 1004, False),
}
FULL_BODIES = {
    'full': dict(rings=84, segs=82),
    'full2': dict(rings=84, segs=82),
    'ico_full': dict(topology='ico', freq=26),
}

if __name__ == '__main__':
    which = sys.argv[1:] or (list(BODIES) + [t + s for t in FULL_BODIES for s in ('', '_train')])
    for tag, (kw, batch, seed, dense) in BODIES.items():
        if tag in which:
            contact_case(tag, kw, batch=batch, seed=seed, store_dense=dense)
    for tag in FULL_BODIES:
        if tag in which:
            fullsize_case(tag)
        if tag + '_train' in which:
            fullsize_train_case(tag)
