"""Which torch ops does one TUCH.forward_train_step (+ backward) launch?  (torch.profiler; config-4 shard shape)"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
p = bench.build_problem(32, dev, 1004)
fn = bench.make_tuch_step(p, run_smplify=False)
for _ in range(3):
    fn()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cpu_time_total', row_limit=45, max_name_column_width=60))
