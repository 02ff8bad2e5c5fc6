"""How full are the tiles of the HD nearest-point search (hd_search_kernel) that reach the matrix-core products?  Builds a
second library with -DTUCH_SCAN_COUNTS for hd_search.hip, runs one HD contact loss at batch 64, prints tiles and columns.

    python tools/diag/hd_counts.py build      # here
    python tools/diag/hd_counts.py            # on the GPU box
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tuch_amd', 'libtuch_amd_hdcounts.so')
if len(sys.argv) > 1 and sys.argv[1] == 'build':
    from tuch_amd import _build
    _build.build()
    objs = [os.path.join(_build.HERE, 'build', os.path.basename(s)[:-4] + '.o') for s in _build.sources() if not s.endswith('hd_search.hip')]
    obj = os.path.join(_build.HERE, 'build', 'hd_search_counts.o')
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', _build.CSRC, '-mllvm',
                    '-amdgpu-mfma-vgpr-form', '-DTUCH_SCAN_COUNTS', '-c', os.path.join(_build.CSRC, 'hd_search.hip'), '-o', obj], check=True)
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + [obj], check=True)
    print(LIB)
    sys.exit(0)
os.environ['TUCH_AMD_LIB'] = LIB
import ctypes
import torch, bench
from tuch_amd import _C
dev = torch.device('cuda:0')
B = 64
p = bench.build_problem(B, dev, 1002)
crit = bench.regressor_loss(p, True)
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices
valid = torch.ones(B, dtype=torch.bool, device=dev)
for _ in range(2):
    crit.contact_loss(verts, valid)
torch.cuda.synchronize()
L = _C.lib()
L.tuch_debug_hd_counts.argtypes = [ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * 16)()
L.tuch_debug_hd_counts(None, 1)
crit.contact_loss(verts, valid)
torch.cuda.synchronize()
L.tuch_debug_hd_counts(out, 0)
c = list(out)
print('tiles reaching the products: %d (%.0f per body); columns in reach with an admissible run: %.1f of 64 (in reach at all: %.1f)'
      % (c[0], c[0] / B, c[1] / max(c[0], 1), c[2] / max(c[0], 1)))
print('histogram of live columns per tile (0-7, 8-15, ..., 64):', [round(x / max(c[0], 1), 3) for x in c[3:12]])
