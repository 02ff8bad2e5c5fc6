"""Which torch-side kernels does one eager stage-2 step launch, and from where?  (torch.profiler, a few steps)"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
p = bench.build_problem(64, dev, 1002)
fn = bench.make_step(p)
for _ in range(3):
    fn()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cpu_time_total', row_limit=60, max_name_column_width=60))
