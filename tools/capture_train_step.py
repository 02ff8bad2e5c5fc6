import os, sys; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench, time
dev = torch.device('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream(dev))
p = bench.build_problem(32, dev, 1004)
fn = bench.make_tuch_step(p, run_smplify=False)
t = bench.time_kernel(fn, 10); print('eager ms', t * 1e3)
try:
    g = bench.capture(fn, 3)
    t = bench.time_kernel(g, 20); print('graph ms', t * 1e3)
except Exception as e:
    import traceback; traceback.print_exc()
