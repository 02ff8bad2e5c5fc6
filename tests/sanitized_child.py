"""Runs inside a python whose libtuch_amd is the ASan + UBSan build (tests/test_sanitized_host.py sets LD_PRELOAD and
TUCH_AMD_LIB): exercises the host-side table builders.  Any sanitizer report aborts the process (halt_on_error)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np


def host_part():
    """csrc/cluster_tree.hip: the tree builder (farthest-point clusters, ear removal, pairwise merge, caps, strips,
    frontiers, launch order) on regular and irregular topologies, several leaf sizes, and the rejected inputs."""
    from helpers import golden
    from tuch_amd import _C, ops
    assert 'asan' in _C.LIB_PATH, _C.LIB_PATH
    for tag, leaves in (('small', (8, 16, 64)), ('ico_small', (8, 16)), ('medium', (16, 32, 64)), ('ico_medium', (24, 32)),
                        ('full', (32,)), ('ico_full', (32, 64))):
        g = golden(tag)
        nv = int(g['faces'].max()) + 1
        for leaf in leaves:
            t = ops.cluster_tree(g['faces'], nv, leaf_faces=leaf)
            nodes = t['nodes']
            assert nodes.shape[0] >= 1 and t['exact_len'] > 0
            again = ops.cluster_tree(g['faces'], nv, leaf_faces=leaf)
            assert all(np.array_equal(t[k], again[k]) for k in t if isinstance(t[k], np.ndarray))
    g = golden('small')
    nv = int(g['faces'].max()) + 1
    for bad in (g['faces'][:-1], np.concatenate([g['faces'][:-1], g['faces'][-1:, ::-1]])):     # open / inconsistent
        try:
            ops.cluster_tree(bad, nv, leaf_faces=16)
        except _C.TuchError:
            pass
        else:
            raise AssertionError('an open or inconsistently oriented mesh was accepted')


def tables_part():
    """csrc/model.hip (tuch_contact_model_create: strips, rings, cluster tree + mask tables, segment assist tables, region
    tables, mask packing), csrc/hd_contact.hip (tuch_hd_model_create: sorted point tables, the adjoint's CSR) and
    csrc/smpl_lbs.hip (tuch_smpl_model_create: the folded joint regressor) with TUCH_HOST_TABLES=1: the finished tables
    stay in host memory instead of being uploaded, so the builders run -- instrumented -- without a device."""
    import torch
    import golden_io as gio
    from helpers import golden, golden_mask
    from tuch_amd import lbs
    from tuch_amd.models.smpl import SMPL
    from tuch_amd.ops import ContactModel, HDModel
    from synthetic import make_body
    assert os.environ.get('TUCH_HOST_TABLES') == '1'
    dev = torch.device('cpu')
    for tag in ('small', 'ico_small', 'medium', 'ico_medium', 'full', 'ico_full'):
        g = golden(tag)
        segs = gio.unpack_segments(g)
        regions, pairs = gio.unpack_regions(g)
        names = list(regions.keys())
        pair_idx = np.asarray([[names.index(a), names.index(b)] for a, b in pairs], np.int64)
        for with_mask in (True, False):
            model = ContactModel(g['faces'], golden_mask(tag) if with_mask else None,
                                 [(s['vidx'], list(s['bands'].values())) for s in segs.values()],
                                 [regions[n] for n in names], pair_idx if with_mask else None, device=dev)
            assert model._handle
            vidx, sign, n_strips = model.strips()
            assert len(vidx) == len(sign) and n_strips > 0
            if with_mask and 'hd_idx' in g:
                hd = HDModel(model, g['hd_idx'], g['hd_w'], g['hd_face'])
                assert hd._handle
                del hd
            del model
    body = make_body(20, 20, seed=3)
    model = ContactModel(body.faces, body.geodesics > 0.3, [(s['vidx'], list(s['bands'].values())) for s in body.segments.values()],
                         device=dev)
    hd = HDModel(model, body.hd_bary_idx, body.hd_bary_w, body.hd_face_id)
    assert hd._handle
    del hd, model
    smpl = SMPL(model_data=body, batch_size=2)
    dm = lbs.SmplDeviceModel(smpl, dev)
    assert dm._handle
    del dm


if __name__ == '__main__':
    host_part()
    tables_part()
    print('SANITIZED-OK', flush=True)
