"""train.py-style contact step, plain and HD, eager and replayed (bench.contact_loss_eval's figures alone)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = bench.build_problem(B, dev, 1002)
for hd in (False, True):
    eager = bench.time_kernel(bench.make_train_step(p, hd), 5) * 1e3
    graph = bench.time_kernel(bench.capture(bench.make_train_step(p, hd), 3), 10) * 1e3
    print('B=%d %s: eager %.4f ms, graph %.4f ms' % (B, 'hd' if hd else 'plain', eager, graph), flush=True)
