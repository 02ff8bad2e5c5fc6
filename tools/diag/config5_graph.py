"""BASELINE config 5 per-rank step (TUCH.forward_train_step --run_smplify, 64 bodies): eager vs ONE captured hipGraph."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = bench.build_problem(B, dev, 1004)
step = bench.make_tuch_step(p, run_smplify=True, smplify_iters=10)
print('eager: %.3f ms' % (bench.time_kernel(step, 2) * 1e3), flush=True)
stats = step()
print('loss eager', float(stats[0]), flush=True)
g = bench.capture(step, 3)
print('graph: %.3f ms' % (bench.time_kernel(g, 5) * 1e3), flush=True)
print('loss replayed', float(g()[0]), flush=True)
