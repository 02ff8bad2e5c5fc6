"""Which kernels (ours and torch's) one train.py-style contact step launches, in order (eager, torch.profiler):
    python tools/diag/train_ops.py [hd]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
use_hd = len(sys.argv) > 1 and sys.argv[1] == 'hd'
p = bench.build_problem(64, dev, 1002)
fn = bench.make_train_step(p, use_hd)
for _ in range(3):
    fn()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    fn()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
evs.sort(key=lambda e: e.time_range.start)
t0 = evs[0].time_range.start
for e in evs:
    print('%8.1f %7.1f  %s' % (e.time_range.start - t0, e.time_range.end - e.time_range.start, e.name[:110]))
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=40, max_name_column_width=50))
