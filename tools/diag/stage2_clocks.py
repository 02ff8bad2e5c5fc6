"""Where the stage-2 tail kernel's time goes: a second library with -DTUCH_STAGE2_CLOCKS stamps s_memrealtime (100 MHz) in
block (0, 0) and in the block that arrives last.   python tools/diag/stage2_clocks.py build | [batch]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tuch_amd', 'libtuch_amd_clocks.so')
if len(sys.argv) > 1 and sys.argv[1] == 'build':
    from tuch_amd import _build
    _build.build()
    objs = [os.path.join(_build.HERE, 'build', os.path.basename(s)[:-4] + '.o') for s in _build.sources()
            if not s.endswith('contact_terms.hip')]
    obj = os.path.join(_build.HERE, 'build', 'contact_terms_clocks.o')
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', _build.CSRC,
                    '-DTUCH_STAGE2_CLOCKS', '-c', os.path.join(_build.CSRC, 'contact_terms.hip'), '-o', obj], check=True)
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + [obj], check=True)
    print(LIB)
    sys.exit(0)
os.environ['TUCH_AMD_LIB'] = LIB
import ctypes, torch, bench
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
L.tuch_debug_stage2_clocks.argtypes = [ctypes.c_void_p]
out = (ctypes.c_ulonglong * 16)()
L.tuch_debug_stage2_clocks(out)
c = list(out)
us = lambda a, b: (c[b] - c[a]) / 100.0
print('block (0,0): loads %.2f us, terms + atomics %.2f, r2r part %.2f, block sums %.2f, ticket %.2f' %
      (us(0, 1), us(1, 2), us(2, 7), us(7, 3), us(3, 4)))
print('last block: arrives %.2f us after block (0,0) started; final reduction %.2f us' % (us(0, 5), us(5, 6)))
