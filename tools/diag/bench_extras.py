import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench
dev = torch.device('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream(dev))
what = sys.argv[1]
if what == 'shard': print(bench.shard_sweep(dev, 1002))
elif what == 'workloads': print(bench.workloads(dev, 1002))
elif what == 'worst': print(bench.worst_case(dev, 1002, 64))
elif what == 'folded': print(bench.worst_case(dev, 1002, 64, folded=True))
elif what == 'irregular': print(bench.irregular_topology(dev, 1002, 64))
elif what == 'nohints': print(bench.headline_without_hints(dev, 1002, 64))
