import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device('cuda:0')
p = bench.build_problem(64, dev, 1002)
fn = bench.make_train_step(p, True)
for _ in range(4):
    fn()
torch.cuda.synchronize()
