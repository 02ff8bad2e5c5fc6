"""Loaders for the reference's asset files, so that ``train.py`` /
``demo_smplify_dc.py`` can run against this package on a machine that has the licensed data.
None of these files ship; the loaders are exercised on synthetic files of the same formats.

    configs/config.py:74-87   SMPL_MODEL_DIR, GEODESICS_SMPL, SEGMENT_DIR, HD_MODEL_DIR, DSC_ROOT
"""
from __future__ import annotations

import os
import pickle
import struct
from typing import Dict, Tuple

import numpy as np


# configs/config.py:74-92 -- used when the reference checkout (and with it ``configs.config``) is not importable
_DEFAULT_PATHS = {
    'SMPL_MODEL_DIR': 'data/models/smpl',
    'JOINT_REGRESSOR_TRAIN_EXTRA': 'data/essentials/spin/J_regressor_extra.npy',
    'PRIOR_FOLDER': 'data/essentials/spin',
    'GEODESICS_SMPL': 'data/essentials/geodesics/smpl/smpl_neutral_geodesic_dist.npy',
    'HD_MODEL_DIR': 'data/essentials/hd_model/smpl',
    'SEGMENT_DIR': 'data/essentials/segments/smpl',
    'STATIC_FITS_DIR': 'data/static_fits',
    'DSC_ROOT': '//tuch/dsc/release',
}


def config_path(name: str) -> str:
    """``configs.config.<name>`` of the reference checkout when importable, else its shipped default."""
    try:
        import importlib
        cfg = importlib.import_module('configs.config')
        if hasattr(cfg, name):
            return getattr(cfg, name)
    except ImportError:
        pass
    return _DEFAULT_PATHS[name]


def reference_segm_utils():
    """``data.essentials.segments.smpl.segm_utils`` (segmentation.py:26: boundary loops per segment); part of the
    licensed data folder.  Raises ImportError with a pointer when it is not importable."""
    import importlib
    try:
        return importlib.import_module('data.essentials.segments.smpl.segm_utils')
    except ImportError as exc:
        raise ImportError('the body-segment tables (data/essentials/segments/smpl/segm_utils.py) are not importable: '
                          'run from a directory that holds the reference\'s data/ folder, or pass the segment '
                          'tables explicitly (segments=...)') from exc


def load_geodesic_mask(path: str, geothres: float) -> np.ndarray:
    """smpl_neutral_geodesic_dist.npy [V,V] -> bool mask geod > geothres (smplifydc.py:65, loss.py:71)."""
    geod = np.load(path)
    return geod > geothres


def load_contact_regions(dsc_root: str = None) -> Dict[str, object]:
    """classes.pkl + ContactSigSMPL.pkl -> {'classes': [...], 'csig': {...}} (train_module.py:64-66)."""
    if dsc_root is None:
        dsc_root = config_path('DSC_ROOT')
    with open(os.path.join(dsc_root, 'classes.pkl'), 'rb') as f:
        classes = pickle.load(f)
    with open(os.path.join(dsc_root, 'ContactSigSMPL.pkl'), 'rb') as f:
        csig = pickle.load(f)
    return {'classes': classes, 'csig': csig}


def sparse_rows(dense: np.ndarray, max_nnz: int = 8, negligible: float = 1e-7, return_dropped: bool = False):
    """Dense regressor [N,V] -> (ids [N,K], weights [N,K]) with K = the largest number of non-zeros in a row (at least 3:
    barycentric samples; rows with fewer are padded with weight 0).  The reference multiplies the dense matrix
    (loss.py:285); the device tables hold its non-zeros -- EVERY non-zero, whatever its size, as long as no row has more
    than ``max_nnz`` of them.  Only a matrix that does not fit is thinned: entries below ``negligible`` x their row's largest
    magnitude are then dropped (a float32 sum of the row cannot see them), with a warning; a row that still has more than
    ``max_nnz`` is refused.  return_dropped: also the largest absolute row sum of dropped entries (0.0: exact)."""
    mag = np.abs(dense)
    keep = mag > 0
    nnz = int(keep.sum(1).max()) if dense.size else 0
    dropped = 0.0
    if nnz > max_nnz:
        keep = mag > negligible * mag.max(1, keepdims=True)
        nnz = int(keep.sum(1).max())
        if nnz > max_nnz:
            raise ValueError('HD regressor rows have up to %d non-zeros (at most %d are supported)' % (nnz, max_nnz))
        dropped = float(np.where(keep, 0, mag).sum(1).max())
        import warnings
        warnings.warn('HD regressor: rows with more than %d non-zeros -- entries below %g of their row maximum dropped '
                      '(largest row sum of them: %.3g)' % (max_nnz, negligible, dropped))
    k = max(nnz, 3)
    idx = np.argsort(-np.where(keep, mag, 0), axis=1, kind='stable')[:, :k]
    wgt = np.where(np.take_along_axis(keep, idx, 1), np.take_along_axis(dense, idx, 1), 0).astype(np.float32)
    idx = np.where(wgt != 0, idx, idx[:, :1])               # padding: weight 0, a valid id
    if return_dropped:
        return idx.astype(np.int64), wgt, dropped
    return idx.astype(np.int64), wgt


def load_hd_regressor(hd_model_dir: str = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """smpl_neutral_hd_vert_regressor.npy (dense [N,V]; barycentric samples: <= 3 non-zeros per row, a general sparse
    regressor with up to 8 also loads) and smpl_neutral_hd_sample_from_mesh_out.pkl -> (vertex ids [N,K], weights [N,K],
    face ids [N]) (loss.py:81-88)."""
    if hd_model_dir is None:
        hd_model_dir = config_path('HD_MODEL_DIR')
    dense = np.load(os.path.join(hd_model_dir, 'smpl_neutral_hd_vert_regressor.npy'))
    idx, wgt = sparse_rows(dense)
    with open(os.path.join(hd_model_dir, 'smpl_neutral_hd_sample_from_mesh_out.pkl'), 'rb') as f:
        faces = np.asarray(pickle.load(f)['faces_vert_is_sampled_from'])
    return idx, wgt, faces.astype(np.int64)


_PLY_TYPES = {'char': 'b', 'int8': 'b', 'uchar': 'B', 'uint8': 'B', 'short': 'h', 'int16': 'h',
              'ushort': 'H', 'uint16': 'H', 'int': 'i', 'int32': 'i', 'uint': 'I', 'uint32': 'I',
              'float': 'f', 'float32': 'f', 'double': 'd', 'float64': 'd'}


def read_ply_vertex_red(path: str) -> np.ndarray:
    """The red channel of every vertex of an ascii or binary PLY file -- all the reference needs from
    the segment meshes (segmentation.py:40-42, `visual.vertex_colors[:, 0] == 255`), without trimesh."""
    with open(path, 'rb') as f:
        if f.readline().strip() != b'ply':
            raise ValueError('%s is not a PLY file' % path)
        fmt, n_vert, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline().decode('ascii', 'replace').strip()
            tok = line.split()
            if not tok:
                continue
            if tok[0] == 'format':
                fmt = tok[1]
            elif tok[0] == 'element':
                in_vertex = tok[1] == 'vertex'
                if in_vertex:
                    n_vert = int(tok[2])
            elif tok[0] == 'property' and in_vertex:
                if tok[1] == 'list':
                    raise ValueError('list property on vertices is not supported')
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == 'end_header':
                break
        names = [p[0] for p in props]
        if 'red' not in names:
            raise ValueError('%s has no per-vertex colours' % path)
        red_i = names.index('red')
        if fmt == 'ascii':
            rows = [f.readline().split() for _ in range(n_vert)]
            return np.asarray([float(r[red_i]) for r in rows]).astype(np.int64)
        endian = '<' if fmt == 'binary_little_endian' else '>'
        rec = struct.Struct(endian + ''.join(p[1] for p in props))
        data = f.read(rec.size * n_vert)
        return np.asarray([rec.unpack_from(data, i * rec.size)[red_i] for i in range(n_vert)]).astype(np.int64)


def load_segments(segment_dir: str = None, segm_utils_segments: Dict[str, Dict[str, list]] = None) -> Dict[str, dict]:
    """smpl_segment_{name}.ply + segm_utils.segments -> {name: {'vidx', 'bands'}}, the form
    tuch_amd.utils.segmentation.BatchBodySegment takes (segmentation.py:40-46)."""
    if segment_dir is None:
        segment_dir = config_path('SEGMENT_DIR')
    if segm_utils_segments is None:
        segm_utils_segments = reference_segm_utils().segments
    out = {}
    for name, bands in segm_utils_segments.items():
        red = read_ply_vertex_red(os.path.join(segment_dir, 'smpl_segment_{}.ply'.format(name)))
        out[name] = {'vidx': np.where(red == 255)[0].astype(np.int64),
                     'bands': {k: np.asarray(v, np.int64) for k, v in bands.items()}}
    return out
