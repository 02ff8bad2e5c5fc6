"""Pack / unpack the ragged inputs of the golden fixtures (regions, segments) into
flat arrays an .npz can hold.  Shared by tests/golden/make_golden.py and the tests."""
from __future__ import annotations

import os
from typing import Dict, List

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def pack_ragged(prefix: str, lists: List[np.ndarray], out: dict) -> None:
    out[prefix + '_flat'] = np.concatenate([np.asarray(a, np.int64) for a in lists]) if lists \
        else np.zeros(0, np.int64)
    out[prefix + '_off'] = np.cumsum([0] + [len(a) for a in lists]).astype(np.int64)


def unpack_ragged(prefix: str, data) -> List[np.ndarray]:
    flat, off = data[prefix + '_flat'], data[prefix + '_off']
    return [flat[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def pack_segments(segments: Dict[str, dict], out: dict) -> None:
    names = list(segments.keys())
    out['seg_names'] = np.asarray(names)
    pack_ragged('seg_vidx', [segments[n]['vidx'] for n in names], out)
    bands, owner = [], []
    for i, n in enumerate(names):
        for b in segments[n]['bands'].values():
            bands.append(b)
            owner.append(i)
    pack_ragged('seg_band', bands, out)
    out['seg_band_owner'] = np.asarray(owner, np.int64)


def unpack_segments(data) -> Dict[str, dict]:
    names = [str(n) for n in data['seg_names']]
    vidx = unpack_ragged('seg_vidx', data)
    bands = unpack_ragged('seg_band', data)
    owner = data['seg_band_owner']
    out = {}
    for i, n in enumerate(names):
        mine = [bands[k] for k in range(len(bands)) if owner[k] == i]
        out[n] = {'vidx': vidx[i], 'bands': {'band%d' % k: b for k, b in enumerate(mine)}}
    return out


def pack_regions(regions: Dict[str, np.ndarray], pairs, out: dict) -> None:
    names = list(regions.keys())
    out['region_names'] = np.asarray(names)
    pack_ragged('region', [regions[n] for n in names], out)
    out['region_pairs'] = np.asarray([[names.index(a), names.index(b)] for a, b in pairs], np.int64)


def unpack_regions(data):
    names = [str(n) for n in data['region_names']]
    lists = unpack_ragged('region', data)
    regions = {n: l for n, l in zip(names, lists)}
    pairs = [(names[a], names[b]) for a, b in data['region_pairs']]
    return regions, pairs


def pack_mask(mask: np.ndarray, out: dict) -> None:
    out['geomask_bits'] = np.packbits(np.asarray(mask, bool), axis=None)
    out['geomask_n'] = np.int64(mask.shape[0])


def unpack_mask(data) -> np.ndarray:
    n = int(data['geomask_n'])
    return np.unpackbits(data['geomask_bits'], count=n * n).reshape(n, n).astype(bool)


def load(name: str):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)
