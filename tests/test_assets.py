"""CPU: the loaders for the reference's asset formats, on synthetic files of those formats."""
import os
import pickle
import struct

import numpy as np

from tuch_amd import assets
from tuch_amd.synthetic import dense_hd_regressor, make_body


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
