"""Run one of the bench workloads a few times eagerly (for rocprofv3 --kernel-trace / --pmc).
   python tools/profile_step.py {step|hd|plain|fit} [batch] [iters]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
what = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device('cuda:0')
p = bench.build_problem(B, dev, 1002)
if what == 'step':
    fn = bench.make_step(p)
elif what in ('hd', 'plain'):
    fn = bench.make_train_step(p, what == 'hd')
else:
    fn = bench.make_fit(p, 20)[0]
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    fn()
e1.record()
torch.cuda.synchronize()
print('%s B=%d: %.4f ms per call (eager)' % (what, B, e0.elapsed_time(e1) / iters))
