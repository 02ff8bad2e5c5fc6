"""Loss trajectory of the stage-2 step: eager launches vs hipGraph replay, at several batch sizes."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
for B in [int(x) for x in (sys.argv[1:] or ['4', '8', '16', '64'])]:
    p = bench.build_problem(B, dev, 1002)
    step = bench.make_step(p)
    eager = [float(step()[0]) for _ in range(10)]
    p = bench.build_problem(B, dev, 1002)
    replay = bench.capture(bench.make_step(p), 3)
    graph = [float(replay()[0]) for _ in range(7)]
    print('B=%d eager' % B, ['%.1f' % x for x in eager])
    print('B=%d graph' % B, ['%.1f' % x for x in graph], '(first = eager iteration 4)')
