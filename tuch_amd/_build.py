"""Build libtuch_amd.so (hand-written HIP for gfx950 + the C ABI) in-tree with hipcc.

    python -m tuch_amd._build            # build if stale
    python -m tuch_amd._build --force

The library links only the HIP runtime: no torch types cross the ABI
(include/tuch_amd.h).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libtuch_amd.so')
ARCH = 'gfx950'


# per-file flags.  hd_search.hip, v2v.hip: matrix-core results in ordinary vector registers (the accumulators are read by
# the vector unit at once: as AGPRs every value costs a v_accvgpr_read)
EXTRA_FLAGS = {'hd_search.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form'], 'v2v.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form']}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, 'build', os.path.basename(src)[:-4] + '.o')
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
                os.path.getmtime(src), *[os.path.getmtime(os.path.join(CSRC, h))
                                         for h in os.listdir(CSRC) if h.endswith('.h')]):
            cmd = [hipcc, '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-I', CSRC,
                   '-Wall', '-Wno-unused-function'] + EXTRA_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd))
            subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


ASAN_LIB = os.path.join(HERE, 'libtuch_amd_asan.so')


def sanitizer_runtime() -> str:
    """clang's shared ASan runtime (it carries the UBSan handlers too): LD_PRELOAD it into the python that loads ASAN_LIB."""
    import glob
    hits = sorted(glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so'))
    if not hits:
        raise FileNotFoundError('libclang_rt.asan-x86_64.so not found under /opt/rocm/lib/llvm')
    return hits[-1]


def build_sanitized(force: bool = False) -> str:
    """The same sources with AddressSanitizer + UndefinedBehaviorSanitizer on the HOST side (-fno-gpu-sanitize: device
    code as in the product build) -> libtuch_amd_asan.so.  For the ~1 200 lines of host C++ that build tables (cluster
    tree, strips, rings, segment / region / HD tables): tests/test_sanitized_host.py, tests/test_gpu_hardening.py."""
    from concurrent.futures import ThreadPoolExecutor
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    if not force and os.path.exists(ASAN_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(ASAN_LIB) for d in deps):
        return ASAN_LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    out = os.path.join(HERE, 'build', 'asan')
    os.makedirs(out, exist_ok=True)
    san = ['-fsanitize=address,undefined', '-fno-gpu-sanitize', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer']

    def compile_one(src):
        obj = os.path.join(out, os.path.basename(src)[:-4] + '.o')
        subprocess.run([hipcc, '--offload-arch=' + ARCH, '-O1', '-g', '-std=c++17', '-fPIC', '-I', CSRC] + san
                       + EXTRA_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', obj], check=True)
        return obj
    with ThreadPoolExecutor(max_workers=4) as pool:
        objs = list(pool.map(compile_one, sources()))
    subprocess.run([hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-fsanitize=address,undefined', '-shared-libsan',
                    '-o', ASAN_LIB] + objs, check=True)
    return ASAN_LIB


if __name__ == '__main__':
    if '--sanitized' in sys.argv:
        print(build_sanitized(force='--force' in sys.argv))
    else:
        print(build(force='--force' in sys.argv, verbose=True))
