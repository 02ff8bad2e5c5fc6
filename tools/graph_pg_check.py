"""Does a hipGraph replay of the stage-2 step survive (a) a second process on the same GPU, (b) a process group?
   python tools/graph_pg_check.py B mode     mode: plain | pg-after | pg-before | eager
Run under torch.distributed.run for 2 ranks (both on cuda:0, gloo)."""
import os, sys; sys.path.insert(0, '.')
import torch
import bench
B, mode = int(sys.argv[1]), sys.argv[2]
rank = int(os.environ.get('RANK', '0'))
world = int(os.environ.get('WORLD_SIZE', '1'))
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
if mode == 'pg-before' and world > 1:
    torch.distributed.init_process_group('gloo')
p = bench.build_problem(B, dev, 1002)
step = bench.make_step(p)
if mode != 'eager':
    step = bench.capture(step, 3)
if mode == 'pg-after' and world > 1:
    torch.distributed.init_process_group('gloo')
out = []
for i in range(10):
    s = step()
    if world > 1 and mode.startswith('pg'):
        h = s.cpu(); torch.distributed.all_reduce(h)
    out.append(float(s[0]))
torch.cuda.synchronize()
print('rank %d B=%d %s:' % (rank, B, mode), ['%.4g' % x for x in out], flush=True)
