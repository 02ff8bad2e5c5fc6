"""Timeline of ONE replayed stage-2 step: python tools/graph_timeline.py run [batch | hd]  -> runs warm-up + a few graph replays;
python tools/graph_timeline.py show <results.db>  -> per-kernel start offset / duration of the last replay, the idle gaps
and how much of the step has 1 / 2 kernels in flight.   (rocprofv3 --kernel-trace -d DIR -o kt -- python tools/graph_timeline.py run)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch, bench
    dev = torch.device('cuda:0')
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    if len(sys.argv) > 2 and sys.argv[2] == 'hd':            # the train.py-style HD contact step, eager launches
        p = bench.build_problem(64, dev, 1002)
        fn = bench.make_train_step(p, True)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        return
    if len(sys.argv) > 2 and sys.argv[2] in ('hdg', 'plaing'):  # the train.py-style contact step captured and replayed
        p = bench.build_problem(64, dev, 1002)
        fn = bench.capture(bench.make_train_step(p, sys.argv[2] == 'hdg'), 3)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        return
    p = bench.build_problem(int(sys.argv[2]) if len(sys.argv) > 2 else 64, dev, 1002)
    fn = bench.capture(bench.make_step(p), 3)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()


def show(path):
    import sqlite3
    c = sqlite3.connect(path)
    cols = [d[1] for d in c.execute("pragma table_info('kernels')")]
    rows = list(c.execute("select name, start, end, duration from kernels order by start"))
    # the last replay = the last block of kernels after the largest gap structure: take the final N kernels where N =
    # number of kernels between two occurrences of the first kernel name of the tail
    names = [r[0] for r in rows]
    last = len(rows) - 1
    first_name = None
    # find the period: the name sequence repeats; locate the previous occurrence of the final kernel's name
    end_name = names[-1]
    prev = max(i for i in range(len(names) - 1) if names[i] == end_name)
    step = rows[prev + 1:]
    t0 = step[0][1]
    wall = step[-1][2] - t0
    busy = sum(r[3] for r in step)
    events = sorted([(r[1], 1) for r in step] + [(r[2], -1) for r in step])
    depth, last_t, by_depth = 0, t0, {}
    for t, d in events:
        by_depth[depth] = by_depth.get(depth, 0) + (t - last_t)
        depth += d
        last_t = t
    print('kernels in the step: %d, wall %.1f us, summed kernel time %.1f us' % (len(step), wall / 1e3, busy / 1e3))
    print('time with k kernels in flight:', {k: round(v / 1e3, 1) for k, v in sorted(by_depth.items())})
    for r in step:
        print('%9.1f %8.1f  %s' % ((r[1] - t0) / 1e3, r[3] / 1e3, r[0][:90]))


if __name__ == '__main__':
    run() if sys.argv[1] == 'run' else show(sys.argv[2])
