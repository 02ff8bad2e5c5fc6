"""The host-side C++ of the library (cluster-tree builder, strip / ring / segment / region / HD table construction:
~1 200 lines that run once per model) under AddressSanitizer + UndefinedBehaviorSanitizer.

`tuch_amd._build.build_sanitized()` compiles the same sources with -fsanitize=address,undefined on the host side
(device code as in the product build); a child python preloads clang's ASan runtime and binds that library through
TUCH_AMD_LIB, with TUCH_HOST_TABLES=1: the finished model tables stay in host memory instead of being uploaded, so every
builder runs here, without a device.  (With a device the sanitized library cannot run in this image: ROCm's ASan runtime
intercepts hsa_amd_memory_pool_allocate and aborts beside the non-instrumented HIP runtime torch ships.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_sanitized(*args, timeout=900):
    from tuch_amd import _build
    lib = _build.build_sanitized()
    env = dict(os.environ, LD_PRELOAD=_build.sanitizer_runtime(), TUCH_AMD_LIB=lib, TUCH_HOST_TABLES='1',
               # python itself leaks by design; the HIP runtime maps memory where ASan's default layout expects a gap
               ASAN_OPTIONS='detect_leaks=0:halt_on_error=1:abort_on_error=0:protect_shadow_gap=0',
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'sanitized_child.py')] + list(args), env=env,
                         capture_output=True, text=True, timeout=timeout)
    out = res.stdout + res.stderr
    assert res.returncode == 0, out[-4000:]
    assert 'ERROR: AddressSanitizer' not in out and 'runtime error:' not in out, out[-4000:]
    return out


@pytest.mark.skipif(shutil.which('hipcc') is None and not os.path.exists('/opt/rocm/bin/hipcc'), reason='no hipcc')
def test_host_table_builders_under_asan_ubsan():
    out = run_sanitized()
    assert 'SANITIZED-OK' in out, out[-3000:]
