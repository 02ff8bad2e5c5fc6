"""ONE hipGraph holding the stage-2 step of two (or four) independent part-batches, each a serial chain on a stream of its
own (no inner fork): do the latency-bound head / tail of one part overlap the big kernels of the other?
    python tools/diag/split_chains.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from tuch_amd.smplify.losses import contact_model_for
dev = torch.device('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream(dev))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


whole = bench.capture(bench.make_step(bench.build_problem(B, dev, 1002)), 3)
print('one graph, batch %d, search beside the inside test: %.4f ms' % (B, timeit(whole)), flush=True)
p = bench.build_problem(B, dev, 1002)
model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
for parts in (2, 4):
    model.set_option('overlap', 0)
    probs = [bench.build_problem(B // parts, dev, 1002 + k) for k in range(parts)]
    steps = [bench.make_step(q) for q in probs]
    streams = [torch.cuda.Stream(dev) for _ in range(parts - 1)]

    def both():
        cur = torch.cuda.current_stream()
        outs = []
        for k in range(1, parts):
            streams[k - 1].wait_stream(cur)
            with torch.cuda.stream(streams[k - 1]):
                outs.append(steps[k]())
        outs.append(steps[0]())
        for s in streams:
            cur.wait_stream(s)
        return outs[-1]
    g = bench.capture(both, 3)
    print('one graph, %d serial chains of batch %d: %.4f ms' % (parts, B // parts, timeit(g)), flush=True)
    model.set_option('overlap', 1)
    g = bench.capture(both, 3)
    print('one graph, %d chains of batch %d, each with its own side stream: %.4f ms' % (parts, B // parts, timeit(g)), flush=True)
