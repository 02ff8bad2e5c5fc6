"""CPU: the C-ABI library builds, loads, and exports every symbol include/tuch_amd.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from tuch_amd import _C, _build
    _build.build()
    lib = ctypes.CDLL(_C.LIB_PATH)
    header = open(os.path.join(ROOT, 'include', 'tuch_amd.h')).read()
    declared = sorted(set(re.findall(r'\b(tuch_[a-z0-9_]+)\s*\(', header)))
    assert declared, 'no declarations found'
    for name in declared:
        assert hasattr(lib, name), 'missing symbol ' + name
    # the Python binding and the header agree
    assert set(_C.exported_symbols()) <= set(declared), set(_C.exported_symbols()) - set(declared)
    assert _C.lib().tuch_abi_version() >= 1


def test_error_reporting_without_gpu():
    from tuch_amd import _C
    L = _C.lib()
    rc = L.tuch_winding_numbers(None, None, 1, 1, 1, None, None, 0.99, None, 0, None)
    assert rc != 0
    assert b'null pointer' in L.tuch_last_error()
    assert L.tuch_geomask_words(6890) == 108
    assert L.tuch_winding_workspace_bytes(64, 6890, 13776) > 0
