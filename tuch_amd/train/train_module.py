"""The contact helper of the reference's ``tuch/train/train_module.py`` that sits on the hot path:
``TUCH.contact_from_verts`` (train_module.py:69-91, "Speed up this function will speed up
training loop!").  The rest of that class (data loading, SPIN, fits dictionary, rendering) is
out of scope; ``TUCH`` here carries only what ``contact_from_verts`` needs, with the same
method signature, so the reference's training step can call it unchanged."""
from __future__ import annotations

import numpy as np
import torch

from .. import ops


class TUCH:
    def __init__(self, contactlists, faces, device=None):
        """contactlists = {'classes': [(regionA, regionB), ...], 'csig': {region: vertex ids}}
        (classes.pkl / ContactSigSMPL.pkl, train_module.py:64-66)."""
        self.contactlists = contactlists
        self.device = device
        names = list(contactlists['csig'].keys())
        index = {n: i for i, n in enumerate(names)}
        pairs = np.asarray([[index[str(a)], index[str(b)]] for a, b in contactlists['classes']], np.int64)
        self._model = ops.ContactModel(faces, None, None, [np.asarray(contactlists['csig'][n]) for n in names],
                                       pairs, device=device)

    def contact_from_verts(self, verts, mode='regions'):
        """[B,V,3] -> [B,P]: minimum squared distance between the two regions of every pair."""
        if mode != 'regions':
            raise ValueError("only mode='regions' exists in the reference")
        return self._model.region_pair_min(verts)[0]
