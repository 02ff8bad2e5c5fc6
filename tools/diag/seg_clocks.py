"""Where segment_one_kernel's time goes inside ONE replayed step: a second library with -DTUCH_SEG_CLOCKS stamps
s_memrealtime (100 MHz) of every block at the kernel's phase boundaries.  python tools/diag/seg_clocks.py build | [batch]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tuch_amd', 'libtuch_amd_clocks.so')
if len(sys.argv) > 1 and sys.argv[1] == 'build':
    from tuch_amd import _build
    _build.build()
    objs = [os.path.join(_build.HERE, 'build', os.path.basename(s)[:-4] + '.o') for s in _build.sources()
            if not s.endswith('ray_winding.hip')]
    obj = os.path.join(_build.HERE, 'build', 'ray_winding_clocks.o')
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', _build.CSRC,
                    '-DTUCH_SEG_CLOCKS', '-c', os.path.join(_build.CSRC, 'ray_winding.hip'), '-o', obj], check=True)
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + [obj], check=True)
    print(LIB)
    sys.exit(0)
os.environ['TUCH_AMD_LIB'] = LIB
import ctypes, numpy as np, torch, bench
from tuch_amd import _C
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.cuda.set_stream(torch.cuda.Stream(dev))
p = bench.build_problem(B, dev, 1002)
fn = bench.capture(bench.make_step(p), 3)
for _ in range(5):
    fn()
torch.cuda.synchronize()
L = _C.lib()
L.tuch_debug_seg_clocks.argtypes = [ctypes.c_void_p]
out = np.zeros((8192, 8), np.uint64)
L.tuch_debug_seg_clocks(out.ctypes.data_as(ctypes.c_void_p))
nb = 6 * B * 8
c = out[:nb].astype(np.int64)
t0 = c[:, 0].min()
worked = c[:, 4] > 0
print('%d blocks, %d with interior vertices of their own' % (nb, int(worked.sum())))
print('starts: %.2f .. %.2f us' % (0.0, (c[:, 0].max() - t0) / 100.0))
print('compaction done: median %.2f, latest %.2f us' % (np.median(c[:, 1] - t0) / 100.0, (c[:, 1].max() - t0) / 100.0))
if worked.any():
    w = c[worked]
    for i, name in ((2, 'caps'), (3, 'entries walked'), (4, 'end')):
        print('%-16s median %.2f, latest %.2f us after the first start; phase itself median %.2f, longest %.2f us' %
              (name, np.median(w[:, i] - t0) / 100.0, (w[:, i].max() - t0) / 100.0,
               np.median(w[:, i] - w[:, i - 1]) / 100.0, (w[:, i] - w[:, i - 1]).max() / 100.0))
# the blocks that end last: when they started, their phases, their z
order = np.argsort(-c[:nb, 4])[:12]
print('blocks that end last: (index, segment, body, z) start | compaction | caps | entries | end  [us after the first start; phase lengths]')
for i in order:
    r = c[i]
    if r[4] == 0:
        continue
    sgn = 6
    print('  %5d (seg %d, body %2d, z %d)  start %.1f | %.1f | %.1f | %.1f | %.1f   end %.1f' % (
        i, i % sgn, (i // sgn) % B, i // (sgn * B), (r[0] - t0) / 100.0, (r[1] - r[0]) / 100.0, (r[2] - r[1]) / 100.0,
        (r[3] - r[2]) / 100.0, (r[4] - r[3]) / 100.0, (r[4] - t0) / 100.0))
