"""The reference's scripts construct the hot-path objects like this (train.py:57-87,
demo_smplify_dc.py:54-87, fit_eft.py:48-73):

    smpl = SMPL(config.SMPL_MODEL_DIR, batch_size=..., create_transl=False).to(device)
    smplify = SMPLifyDC(step_size=1e-2, batch_size=..., num_iters=..., focal_length=constants.FOCAL_LENGTH,
                        geodistssmpl=geodistssmpl, geothres=config.geothres, euclthres=config.euclthres)
    loss = RegressorLoss(options=options, device=device, num_verts=num_verts, faces=face_tensor,
                         geodistssmpl=geodistssmpl, geothres=config.geothres, face_tensor=face_tensor)
    segments = BatchBodySegment([x for x in exn.segments.keys()], face_tensor[0])

i.e. with nothing but paths from configs.config; every asset is loaded inside the constructors.  These tests
write a synthetic data/ tree in the reference's file formats (synthetic.write_reference_assets), make
it the working directory, map the package onto the reference's module paths (compat.install) and run those
calls as the scripts spell them.  CPU: construction + the loaded tables (no GPU needed: the device copy is made on
first use; SMPLifyDC gets device= because its default is 'cuda').  GPU: the same calls verbatim and then the
losses are evaluated and compared with objects built from injected arrays.
"""
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

from synthetic import make_body, write_reference_assets


class _Tree:
    """A synthetic data/ tree as working directory + on sys.path, with the reference's module names mapped."""

    def __init__(self, root, body):
        self.root, self.body = str(root), body

    def __enter__(self):
        import tuch_amd.compat as compat
        write_reference_assets(self.body, self.root)
        self.cwd = os.getcwd()
        self.saved = {k: v for k, v in sys.modules.items()
                      if k.split('.')[0] in ('tuch', 'configs', 'data', 'torchgeometry')}
        for k in self.saved:
            del sys.modules[k]
        os.chdir(self.root)
        sys.path.insert(0, self.root)
        importlib.invalidate_caches()
        compat.install()
        return self

    def __exit__(self, *exc):
        os.chdir(self.cwd)
        sys.path.remove(self.root)
        for k in [k for k in sys.modules if k.split('.')[0] in ('tuch', 'configs', 'data', 'torchgeometry')]:
            del sys.modules[k]
        sys.modules.update(self.saved)
        importlib.invalidate_caches()


def _construct(device, batch_size, smplify_device=None):
    """The construction part of train.py / demo_smplify_dc.py, spelled as there."""
    from configs import config
    from data.essentials import constants
    from data.essentials.segments.smpl import segm_utils as exn
    from tuch.models.smpl import SMPL
    from tuch.smplify.smplifydc import SMPLifyDC
    from tuch.train.loss import RegressorLoss
    from tuch.utils.segmentation import BatchBodySegment
    options = types.SimpleNamespace(batch_size=batch_size, num_smplify_iters=5, contact_loss_weight=1.0)
    smpl = SMPL(config.SMPL_MODEL_DIR, batch_size=options.batch_size, create_transl=False).to(device)
    num_verts = smpl.get_num_verts()
    face_tensor = torch.tensor(smpl.faces.astype(np.int64), dtype=torch.long, device=device) \
        .unsqueeze_(0).repeat([options.batch_size, 1, 1])
    geodistssmpl = torch.tensor(np.load(config.GEODESICS_SMPL), device=device)
    extra = {} if smplify_device is None else {'device': smplify_device}
    smplify = SMPLifyDC(step_size=1e-2, batch_size=options.batch_size, num_iters=options.num_smplify_iters,
                        focal_length=constants.FOCAL_LENGTH, geodistssmpl=geodistssmpl, geothres=config.geothres,
                        euclthres=config.euclthres, **extra)
    loss = RegressorLoss(options=options, device=device, num_verts=num_verts, faces=face_tensor,
                         geodistssmpl=geodistssmpl, geothres=config.geothres, face_tensor=face_tensor)
    segments = BatchBodySegment([x for x in exn.segments.keys()], face_tensor[0])
    import pickle
    classes = pickle.load(open(os.path.join(config.DSC_ROOT, 'classes.pkl'), 'rb'))
    csig = pickle.load(open(os.path.join(config.DSC_ROOT, 'ContactSigSMPL.pkl'), 'rb'))
    contactlist = {'classes': classes, 'csig': csig}
    return dict(options=options, smpl=smpl, face_tensor=face_tensor, geodistssmpl=geodistssmpl, smplify=smplify,
                loss=loss, segments=segments, contactlist=contactlist)


def test_reference_constructor_calls_load_every_asset(tmp_path):
    body = make_body(10, 12)
    with _Tree(tmp_path, body):
        o = _construct(torch.device('cpu'), batch_size=2, smplify_device=torch.device('cpu'))
        smpl, loss, smplify, segments = o['smpl'], o['loss'], o['smplify'], o['segments']
        # SMPL: arrays of the pickle (official layouts converted), extra regressor + joint map from the data folder
        assert np.array_equal(smpl.faces, body.faces)
        np.testing.assert_allclose(smpl.v_template.numpy(), body.v_template, atol=1e-7)
        np.testing.assert_allclose(smpl.posedirs.numpy(), body.posedirs, atol=1e-7)
        np.testing.assert_allclose(smpl.J_regressor.numpy(), body.J_regressor, atol=1e-7)
        np.testing.assert_allclose(smpl.J_regressor_extra.numpy(), body.J_regressor_extra, atol=1e-7)
        assert np.array_equal(smpl.parents.numpy(), body.parents)
        assert np.array_equal(smpl.joint_map.numpy(), body.joint_map)
        assert np.array_equal(smpl.extra_vertex_ids.numpy(), body.extra_vertex_ids)
        # RegressorLoss: segments always built (loss.py:91), HD tables from HD_MODEL_DIR (loss.py:81-88)
        assert loss.segments is not None and list(loss.segments.names) == list(body.segments.keys())
        assert loss.use_hd and loss.hd_idx.shape == (len(body.hd_face_id), 3)
        assert sorted(loss.geovec.tolist()) == sorted(body.hd_face_id.tolist())
        for name, seg in body.segments.items():
            for s in (segments.segmentation[name], loss.segments.segmentation[name]):
                assert np.array_equal(s.segment_vidx, seg['vidx'])
                assert [list(b) for b in s.bands_verts] == [list(b) for b in seg['bands'].values()]
        # SMPLifyDC: GMM prior from PRIOR_FOLDER, its own SMPL, the ignored joints by name through constants
        assert smplify.ign_joints == [1, 9, 12, 27, 28]
        np.testing.assert_allclose(smplify.pose_prior.means.numpy(), body.gmm['means'], atol=1e-6)
        assert smplify.face_tensor.shape == (2, body.num_faces, 3)
        assert torch.equal(smplify.geomask, o['geodistssmpl'] > 0.3)
        # train_module is mapped too (contact_from_verts carrier)
        from tuch.train.train_module import TUCH
        assert hasattr(TUCH, 'contact_from_verts')


def test_missing_assets_raise_instead_of_skipping_the_segment_filter(tmp_path):
    """Without the data folder the reference's calls must fail loudly (the reference would, too)."""
    import types as _t
    from tuch_amd.train.loss import RegressorLoss
    body = make_body(10, 12)
    saved = {k: v for k, v in sys.modules.items() if k.split('.')[0] in ('configs', 'data')}
    for k in saved:
        del sys.modules[k]
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        face_tensor = torch.tensor(body.faces)[None]
        with pytest.raises((ImportError, OSError)):
            RegressorLoss(_t.SimpleNamespace(contact_loss_weight=1.0), 'cpu', body.num_verts, face_tensor,
                          torch.tensor(body.geodesics), geothres=0.3, face_tensor=face_tensor, use_hd=False)
    finally:
        os.chdir(cwd)
        sys.modules.update(saved)


@pytest.mark.gpu
def test_reference_calls_verbatim_on_the_gpu(tmp_path):
    """The same calls with no extra argument, then used: contact loss of RegressorLoss (segments and HD tables
    loaded from files) equals the one built from injected arrays, and a short SMPLifyDC fit runs."""
    from tuch_amd.train.loss import RegressorLoss as Injected
    from tuch_amd.utils.segmentation import BatchBodySegment as InjectedSegments
    from synthetic import random_poses
    body = make_body(14, 16, relax_iters=40)
    dev = torch.device('cuda:0')
    batch = 3
    with _Tree(tmp_path, body):
        o = _construct(dev, batch_size=batch)
        bp, go, be = random_poses(batch, 21)
        t = lambda a: torch.tensor(a, device=dev)
        out = o['smpl'](global_orient=t(go), body_pose=t(bp), betas=t(be))
        verts = out.vertices.detach().requires_grad_(True)
        valid = torch.ones(batch, dtype=torch.bool, device=dev)
        got = o['loss'].contact_loss(verts, valid)
        got.backward()
        segs = InjectedSegments(list(body.segments.keys()), o['face_tensor'][0], body.segments)
        ref = Injected(o['options'], dev, body.num_verts, o['face_tensor'], o['geodistssmpl'], geothres=0.3,
                       face_tensor=o['face_tensor'], segments=segs, hd_regressor=(body.hd_bary_idx, body.hd_bary_w),
                       hd_faces=body.hd_face_id)
        v2 = out.vertices.detach().requires_grad_(True)
        want = ref.contact_loss(v2, valid)
        want.backward()
        assert abs(got.item() - want.item()) <= 1e-5 * abs(want.item()) + 1e-7
        assert torch.allclose(verts.grad, v2.grad, rtol=1e-4, atol=1e-6 * float(v2.grad.abs().max()))
        kp = torch.cat([torch.randn(batch, 49, 2, device=dev) * 30, torch.rand(batch, 49, 1, device=dev)], 2)
        res = o['smplify'](torch.cat([t(go), t(bp)], 1), t(be), torch.tensor([[0., 0., 20.]], device=dev).repeat(batch, 1),
                           torch.zeros(batch, 2, device=dev), kp, use_contact=True, contactlist=o['contactlist'],
                           gt_contact=[torch.zeros(batch, len(body.region_pairs), device=dev), None],
                           ignore_idxs=torch.zeros(batch, dtype=torch.bool, device=dev),
                           has_discrete_contact=torch.ones(batch, dtype=torch.bool, device=dev),
                           segments=o['loss'].segments)
        assert torch.isfinite(res[0]).all() and len(res[6]) == 5
