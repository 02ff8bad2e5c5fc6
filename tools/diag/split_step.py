"""Does splitting the batch over concurrently replayed graphs help?  python tools/diag/split_step.py [parts]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream(dev))
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 64
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
whole = bench.capture(bench.make_step(bench.build_problem(B, dev, 1002)), 3)
print('one graph, batch %d: %.4f ms' % (B, timeit(whole)))
for parts in (2, 4):
    streams = [torch.cuda.Stream(dev) for _ in range(parts)]
    graphs = []
    for k in range(parts):
        with torch.cuda.stream(streams[k]):
            graphs.append(bench.capture(bench.make_step(bench.build_problem(B // parts, dev, 1002 + k)), 3))
    torch.cuda.synchronize()
    def both():
        cur = torch.cuda.current_stream()
        for k in range(parts):
            streams[k].wait_stream(cur)
            with torch.cuda.stream(streams[k]):
                graphs[k]()
        for k in range(parts):
            cur.wait_stream(streams[k])
    print('%d graphs of batch %d on %d streams: %.4f ms' % (parts, B // parts, parts, timeit(both)))
    print('   the same, one after the other: %.4f ms' % timeit(lambda: [g() for g in graphs]))
