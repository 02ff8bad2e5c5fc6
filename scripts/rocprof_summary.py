#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (ROCm 7.x default output) into a small text table:
per-kernel calls / total / average / min / max duration, plus VGPR/SGPR/LDS per kernel.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o run -- python bench.py ...
    python scripts/rocprof_summary.py gpurun_out/prof/run_results.db > profiles/r01_bench_kernels.txt
"""
import sqlite3
import sys


def main(path, top=30):
    c = sqlite3.connect(path)
    rows = list(c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(grid_y), max(grid_z), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows) or 1
    print('# rocprofv3 --kernel-trace summary of %s' % path)
    print('# durations in microseconds; pct of summed kernel time (%.3f ms)' % (total / 1e6))
    print('%-72s %6s %12s %10s %10s %10s %6s %5s %5s %6s %s' % (
        'kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct', 'vgpr', 'sgpr', 'lds', 'grid(threads)xwg'))
    for r in rows[:top]:
        print('%-72s %6d %12.1f %10.1f %10.1f %10.1f %6.2f %5s %5s %6s %sx%sx%s/%s' % (
            r[0][:72], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total,
            r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
    try:
        pmc = list(c.execute("select * from counters_collection limit 1"))
        if pmc:
            cols = [d[1] for d in c.execute("pragma table_info('counters_collection')")]
            print('\n# counters_collection columns: %s' % ', '.join(cols))
    except sqlite3.Error:
        pass


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
