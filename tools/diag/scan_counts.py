"""Where the nearest-vertex scan's work goes (one eager stage-2 step at batch 64): builds a second library with
-DTUCH_SCAN_COUNTS (csrc/v2v.hip: counters per wavefront-level event), runs the step with it, prints the counts.

    python tools/diag/scan_counts.py build      # here (hipcc cross-compiles): tuch_amd/libtuch_amd_counts.so
    python tools/diag/scan_counts.py            # on the GPU box
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tuch_amd', 'libtuch_amd_counts.so')

if len(sys.argv) > 1 and sys.argv[1] == 'build':
    from tuch_amd import _build
    _build.build()
    objs = [os.path.join(_build.HERE, 'build', os.path.basename(s)[:-4] + '.o') for s in _build.sources()
            if not s.endswith('v2v.hip')]
    obj = os.path.join(_build.HERE, 'build', 'v2v_counts.o')
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', _build.CSRC,
                    '-DTUCH_SCAN_COUNTS', '-c', os.path.join(_build.CSRC, 'v2v.hip'), '-o', obj], check=True)
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + [obj], check=True)
    print(LIB)
    sys.exit(0)

os.environ['TUCH_AMD_LIB'] = LIB
import ctypes
import torch, bench
from tuch_amd import _C
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = bench.build_problem(B, dev, 1002)
fn = bench.make_step(p)
for _ in range(6):
    fn()
torch.cuda.synchronize()
L = _C.lib()
L.tuch_debug_scan_counts.argtypes = [ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * 32)()
L.tuch_debug_scan_counts(None, 1)
fn()
torch.cuda.synchronize()
L.tuch_debug_scan_counts(out, 0)
names = ['wavefronts', 'leaf-per-lane trips', 'candidate leaves (box-box)', 'leaves reaching rows', 'trips of 8 rows',
         '  skipped by mask', '  taking the update branch', 'trips of 4 rows', '  skipped by mask', '  taking the update branch',
         'wavefronts returning at once', 'candidates passing 16-column sub-blocks', 'candidates passing 8-column sub-blocks', 'trips of 8 with all mask words = all ones', 'trips of 8 with all reach lanes allowed']
c = list(out)
for n, v in zip(names, c):
    print('%-32s %12d  per body %10.1f' % (n, v, v / B))
w = max(c[0] - c[10], 1)
print('columns in reach of a leaf whose rows are evaluated: %.1f of 64 on average' % (c[15] / max(c[3], 1)))
print('per live wavefront: trips %.2f candidates %.2f reaching %.2f rows8 %.2f (skipped %.2f, taking %.3f) rows4 %.2f (skipped %.2f)'
      % (c[1] / w, c[2] / w, c[3] / w, c[4] / w, c[5] / w, c[6] / w, c[7] / w, c[8] / w))
# rough VALU estimate per event (read off the ISA: tools/diag/README)

if c[16] or c[18]:
    live = (c[19] + c[17]) / max(64.0 * (c[18] + c[16]), 1.0)
    print('lane pairs: leaves walked row by row for the wavefront %.2f per wavefront (%.1f of 64 columns in reach), flushes of queued '
          'pairs %.2f per wavefront (%.1f of 64 lanes)' % (c[18] / w, c[19] / max(c[18], 1), c[16] / w, c[17] / max(c[16], 1)))
    print('live_lane_fraction %.3f   # row arithmetic: lanes with a (leaf, column) pair in reach / lanes issued (round 4: 12.2 / 64 = 0.19)' % live)
