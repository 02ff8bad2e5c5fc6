"""CPU: the loaders for the reference's asset formats, on synthetic files of those formats."""
import os
import pickle
import struct

import numpy as np
import pytest
import torch

from tuch_amd import assets
from synthetic import dense_hd_regressor, make_body


def _write_ply(path, verts, red, binary):
    header = ['ply', 'format %s 1.0' % ('binary_little_endian' if binary else 'ascii'),
              'element vertex %d' % len(verts), 'property float x', 'property float y', 'property float z',
              'property uchar red', 'property uchar green', 'property uchar blue', 'property uchar alpha',
              'element face 0', 'property list uchar int vertex_indices', 'end_header']
    with open(path, 'wb') as f:
        f.write(('\n'.join(header) + '\n').encode())
        for v, r in zip(verts, red):
            if binary:
                f.write(struct.pack('<fffBBBB', v[0], v[1], v[2], int(r), 0, 0, 255))
            else:
                f.write(('%f %f %f %d 0 0 255\n' % (v[0], v[1], v[2], int(r))).encode())


def test_segment_ply_and_tables_round_trip(tmp_path):
    body = make_body(10, 12, with_geodesics=False)
    segm_utils = {}
    for i, (name, seg) in enumerate(body.segments.items()):
        red = np.zeros(body.num_verts, np.int64)
        red[seg['vidx']] = 255
        _write_ply(os.path.join(tmp_path, 'smpl_segment_%s.ply' % name), body.v_template, red, binary=bool(i % 2))
        segm_utils[name] = {k: [int(x) for x in v] for k, v in seg['bands'].items()}
    got = assets.load_segments(str(tmp_path), segm_utils)
    assert list(got) == list(body.segments)
    for name in got:
        assert np.array_equal(got[name]['vidx'], body.segments[name]['vidx'])
        for k in got[name]['bands']:
            assert np.array_equal(got[name]['bands'][k], body.segments[name]['bands'][k])


def test_geodesics_regions_and_hd_regressor(tmp_path):
    body = make_body(10, 12)
    np.save(os.path.join(tmp_path, 'geod.npy'), body.geodesics)
    assert np.array_equal(assets.load_geodesic_mask(os.path.join(tmp_path, 'geod.npy'), 0.3), body.geodesics > 0.3)
    with open(os.path.join(tmp_path, 'classes.pkl'), 'wb') as f:
        pickle.dump(np.asarray(body.region_pairs), f)
    with open(os.path.join(tmp_path, 'ContactSigSMPL.pkl'), 'wb') as f:
        pickle.dump({k: list(map(int, v)) for k, v in body.regions.items()}, f)
    cd = assets.load_contact_regions(str(tmp_path))
    assert len(cd['classes']) == len(body.region_pairs) and set(cd['csig']) == set(body.regions)
    np.save(os.path.join(tmp_path, 'smpl_neutral_hd_vert_regressor.npy'), dense_hd_regressor(body))
    with open(os.path.join(tmp_path, 'smpl_neutral_hd_sample_from_mesh_out.pkl'), 'wb') as f:
        pickle.dump({'faces_vert_is_sampled_from': body.hd_face_id}, f)
    idx, w, faces = assets.load_hd_regressor(str(tmp_path))
    assert np.array_equal(faces, body.hd_face_id)
    dense = np.zeros((len(idx), body.num_verts), np.float32)
    np.add.at(dense, (np.repeat(np.arange(len(idx)), 3), idx.ravel()), w.ravel())
    assert np.allclose(dense, dense_hd_regressor(body), atol=1e-7)


def test_hd_regressor_with_more_than_three_non_zeros_per_row(tmp_path):
    """The reference multiplies the DENSE regressor (loss.py:285): a file whose rows are not plain barycentric samples
    (here 1 ... 5 non-zeros per row) loads as [N,K] tables with K = the largest row, padded with weight 0; more than 8
    non-zeros in a row are refused."""
    body = make_body(10, 12)
    rng = np.random.default_rng(5)
    n, v = 300, body.num_verts
    dense = np.zeros((n, v), np.float32)
    for r in range(n):
        k = 1 + r % 5
        cols = rng.choice(v, k, replace=False)
        dense[r, cols] = rng.dirichlet(np.ones(k)).astype(np.float32)
    idx, w = assets.sparse_rows(dense)
    assert idx.shape == w.shape == (n, 5)
    back = np.zeros_like(dense)
    np.add.at(back, (np.repeat(np.arange(n), 5), idx.ravel()), w.ravel())
    assert np.array_equal(back, dense)
    assert ((w != 0).sum(1) == 1 + np.arange(n) % 5).all() and (idx >= 0).all() and (idx < v).all()
    # EXACT as long as the rows fit: tiny leftovers are non-zeros like any other (the reference multiplies the dense matrix)
    noisy = dense.copy()
    free = np.flatnonzero(noisy[7] == 0)[:4]
    noisy[7, free] = 1e-12
    idx2, w2, dropped = assets.sparse_rows(noisy, return_dropped=True)
    assert dropped == 0.0 and idx2.shape[1] == 7                    # row 7: 3 weights + 4 leftovers
    back = np.zeros_like(noisy)
    np.add.at(back, (np.repeat(np.arange(n), 7), idx2.ravel()), w2.ravel())
    assert np.array_equal(back, noisy)
    # only a row that does NOT fit is thinned (entries below 1e-7 of the row's largest weight: a float32 row sum cannot see
    # them), with a warning, and what was dropped is returned
    free = np.flatnonzero(noisy[7] == 0)[:8]
    noisy[7, free] = 1e-12
    with pytest.warns(UserWarning):
        idx3, w3, dropped = assets.sparse_rows(noisy, return_dropped=True)
    assert np.array_equal(idx3, idx) and np.array_equal(w3, w)
    assert 0 < dropped <= 12.1e-12
    dense[0, rng.choice(v, 9, replace=False)] = 0.1
    with pytest.raises(ValueError):
        assets.sparse_rows(dense)


def test_smpl_pickle_with_chumpy_typed_arrays_and_sparse_joint_regressor(tmp_path):
    """The official SMPL_*.pkl carries its arrays as ``chumpy.ch.Ch`` objects (state dict with the data under 'x') and
    J_regressor as a scipy.sparse matrix, written by Python 2 (latin1).  Read here without chumpy installed: a stand-in
    module named chumpy exists only while the file is WRITTEN; the loader (models/smpl.py: _read_pickle) must return the
    same arrays as from a plain-numpy pickle, and a body model built from the file the same constants."""
    import sys
    import types
    import scipy.sparse as sp
    from tuch_amd.models import smpl as smpl_mod
    assert 'chumpy' not in sys.modules
    body = make_body(10, 12)
    v = body.num_verts
    ch = types.ModuleType('chumpy.ch')

    class Ch(object):
        def __init__(self, x):
            self.x = np.asarray(x)
            self._dirty_vars = set()           # what real chumpy objects drag along
            self._itr = None

        def __getstate__(self):
            return dict(self.__dict__)
    Ch.__module__ = 'chumpy.ch'
    Ch.__qualname__ = 'Ch'
    ch.Ch = Ch
    pkg = types.ModuleType('chumpy')
    pkg.ch = ch
    kintree = np.stack([np.where(body.parents < 0, 2 ** 32 - 1, body.parents), np.arange(24)]).astype(np.int64)
    model = {'v_template': Ch(body.v_template.astype(np.float64)), 'shapedirs': Ch(body.shapedirs.astype(np.float64)),
             'posedirs': Ch(body.posedirs.reshape(207, v, 3).transpose(1, 2, 0).astype(np.float64)),
             'J_regressor': sp.csc_matrix(body.J_regressor.astype(np.float64)), 'weights': Ch(body.lbs_weights.astype(np.float64)),
             'kintree_table': kintree, 'f': body.faces.astype(np.uint32), 'extra_vertex_ids': body.extra_vertex_ids}
    sys.modules['chumpy'], sys.modules['chumpy.ch'] = pkg, ch
    try:
        with open(os.path.join(tmp_path, 'SMPL_NEUTRAL.pkl'), 'wb') as f:
            pickle.dump(model, f, protocol=2)
    finally:
        del sys.modules['chumpy'], sys.modules['chumpy.ch']
    with pytest.raises(ModuleNotFoundError):
        with open(os.path.join(tmp_path, 'SMPL_NEUTRAL.pkl'), 'rb') as f:
            pickle.load(f, encoding='latin1')                      # plain pickle cannot read it here
    d = smpl_mod._load_model_dir(str(tmp_path))
    assert np.allclose(d['v_template'], body.v_template) and d['posedirs'].shape == (207, 3 * v)
    assert np.allclose(d['posedirs'], body.posedirs.reshape(207, 3 * v), atol=1e-7)
    assert np.allclose(d['J_regressor'], body.J_regressor, atol=1e-7) and np.allclose(d['lbs_weights'], body.lbs_weights)
    assert np.array_equal(d['parents'][1:], body.parents[1:]) and d['parents'][0] == -1
    m = smpl_mod.SMPL(str(tmp_path), batch_size=2, create_transl=False, J_regressor_extra=body.J_regressor_extra,
                      joint_map=body.joint_map)
    assert m.v_template.dtype == torch.float32 and tuple(m.shapedirs.shape) == (v, 3, 10) and m.get_num_verts() == v
