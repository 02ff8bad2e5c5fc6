"""Make this package importable under the reference's module paths.

    import tuch_amd.compat; tuch_amd.compat.install()
    from tuch.utils.contact import winding_numbers          # -> tuch_amd.utils.contact
    from tuch.smplify.smplifydc import SMPLifyDC            # -> tuch_amd.smplify.smplifydc
    from tuch.train.trainer import Trainer                  # -> the reference checkout's own file

Only the modules on the self-contact path are mapped.  Every other `tuch.*` module
(datasets, trainer, hmr, renderer, saver, ...) keeps coming from the reference checkout, wherever on
sys.path it is and whether it was put there before or after install(): the reference's `tuch` is a
namespace package (no __init__.py), and so are the packages install() registers -- their __path__ is
recomputed from sys.path on every lookup, like importlib's own namespace paths.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import os
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
    'tuch.train.fits_dict': 'tuch_amd.train.fits_dict',
    'tuch.eft.loss': 'tuch_amd.eft.loss',
}

_PACKAGES = ('tuch', 'tuch.utils', 'tuch.smplify', 'tuch.models', 'tuch.train', 'tuch.eft')


class _CheckoutPath:
    """__path__ of a registered `tuch...` package: the directories `<entry>/tuch/<sub>` of every
    sys.path entry that has one, looked up again on each use (a checkout added to sys.path after
    install() is seen; nothing is cached).  importlib only iterates a package's __path__."""

    def __init__(self, name):
        self._parts = tuple(name.split('.'))

    def _dirs(self):
        out = []
        for entry in sys.path:
            if not isinstance(entry, str):
                continue
            d = os.path.join(entry or os.getcwd(), *self._parts)
            if os.path.isdir(d) and d not in out:
                out.append(d)
        return out

    def __iter__(self):
        return iter(self._dirs())

    def __len__(self):
        return len(self._dirs())

    def __getitem__(self, i):
        return self._dirs()[i]

    def __contains__(self, item):
        return item in self._dirs()

    def __repr__(self):
        return '_CheckoutPath(%r)' % (self._dirs(),)

    def append(self, item):                 # pkgutil.extend_path and friends: nothing to keep
        pass


class _MappedLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        return importlib.import_module(self.target)

    def exec_module(self, module):          # already executed under its tuch_amd name
        pass


class _MappedFinder(importlib.abc.MetaPathFinder):
    """First on sys.meta_path: the names of _MAP always resolve to the tuch_amd modules, also after
    somebody emptied sys.modules of `tuch.*`; every other name falls through to the ordinary finders."""

    def find_spec(self, fullname, path=None, target=None):
        ours = _MAP.get(fullname)
        if ours is None:
            return None
        return importlib.machinery.ModuleSpec(fullname, _MappedLoader(ours))


_finder = _MappedFinder()


def _package(name):
    """The module registered under `name`: an already imported namespace package of the checkout is
    kept (its path is dynamic as well), anything else is replaced by a namespace-like package."""
    mod = sys.modules.get(name)
    if mod is not None and getattr(mod, '__file__', None) is None and hasattr(mod, '__path__') \
            and not isinstance(mod.__path__, list):
        return mod
    mod = types.ModuleType(name)
    mod.__path__ = _CheckoutPath(name)
    mod.__package__ = name
    sys.modules[name] = mod
    parent, _, leaf = name.rpartition('.')
    if parent:
        setattr(sys.modules[parent], leaf, mod)
    return mod


def install(overwrite: bool = True):
    """Register the tuch_amd modules in sys.modules under the reference's names."""
    if _finder not in sys.meta_path:
        sys.meta_path.insert(0, _finder)
    for pkg in _PACKAGES:
        _package(pkg)
    installed = []
    for ref_name, our_name in _MAP.items():
        if ref_name in sys.modules and not overwrite:
            continue
        mod = importlib.import_module(our_name)
        sys.modules[ref_name] = mod
        parent, _, leaf = ref_name.rpartition('.')
        setattr(sys.modules[parent], leaf, mod)
        installed.append(ref_name)
    importlib.invalidate_caches()
    if install_torchgeometry():
        installed.append('torchgeometry')
    return installed


def uninstall():
    """Undo install(): the finder, the registered packages and the mapped names (tests)."""
    if _finder in sys.meta_path:
        sys.meta_path.remove(_finder)
    for k in [k for k in sys.modules if k == 'tuch' or k.startswith('tuch.')]:
        del sys.modules[k]
    tg = sys.modules.get('torchgeometry')
    if tg is not None and str(getattr(tg, '__version__', '')).endswith('+tuch_amd'):
        del sys.modules['torchgeometry']


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
