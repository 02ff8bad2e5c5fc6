"""Make this package importable under the reference's module paths.

    import tuch_amd.compat; tuch_amd.compat.install()
    from tuch.utils.contact import winding_numbers          # -> tuch_amd.utils.contact
    from tuch.smplify.smplifydc import SMPLifyDC            # -> tuch_amd.smplify.smplifydc

Only the modules on the self-contact path are mapped; everything else of the
reference (datasets, trainer, renderer, ...) keeps coming from the reference checkout, which must be
on sys.path *after* install() has run.
"""
from __future__ import annotations

import importlib
import sys
import types

_MAP = {
    'tuch.utils.contact': 'tuch_amd.utils.contact',
    'tuch.utils.segmentation': 'tuch_amd.utils.segmentation',
    'tuch.utils.geometry': 'tuch_amd.utils.geometry',
    'tuch.smplify.losses': 'tuch_amd.smplify.losses',
    'tuch.smplify.prior': 'tuch_amd.smplify.prior',
    'tuch.smplify.smplifydc': 'tuch_amd.smplify.smplifydc',
    'tuch.models.smpl': 'tuch_amd.models.smpl',
    'tuch.train.loss': 'tuch_amd.train.loss',
    'tuch.train.train_module': 'tuch_amd.train.train_module',
    'tuch.eft.loss': 'tuch_amd.eft.loss',
}


def install(overwrite: bool = True):
    """Register the tuch_amd modules in sys.modules under the reference's names."""
    for pkg in ('tuch', 'tuch.utils', 'tuch.smplify', 'tuch.models', 'tuch.train', 'tuch.eft'):
        if pkg not in sys.modules:
            mod = types.ModuleType(pkg)
            mod.__path__ = []          # namespace-like: lets the reference's other submodules resolve
            sys.modules[pkg] = mod
    installed = []
    for ref_name, our_name in _MAP.items():
        if ref_name in sys.modules and not overwrite:
            continue
        mod = importlib.import_module(our_name)
        sys.modules[ref_name] = mod
        parent, _, leaf = ref_name.rpartition('.')
        setattr(sys.modules[parent], leaf, mod)
        installed.append(ref_name)
    if install_torchgeometry():
        installed.append('torchgeometry')
    return installed


def install_torchgeometry() -> bool:
    """The reference imports two functions from torchgeometry==0.1.2 (train_module.py:27,
    demo_smplify_dc.py:33, fits_dict.py:25).  When that package is absent, a module of that name
    with exactly these two is registered (our device implementations); a real installation is left alone."""
    try:
        importlib.import_module('torchgeometry')
        return False
    except ImportError:
        pass
    from .utils import geometry
    mod = types.ModuleType('torchgeometry')
    mod.rotation_matrix_to_angle_axis = geometry.rotation_matrix_to_angle_axis
    mod.angle_axis_to_rotation_matrix = geometry.angle_axis_to_rotation_matrix
    mod.__version__ = '0.1.2+tuch_amd'
    sys.modules['torchgeometry'] = mod
    return True
