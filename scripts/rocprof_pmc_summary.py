#!/usr/bin/env python3
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES ... --kernel-trace -d out -o run -- <cmd>
    python scripts/rocprof_pmc_summary.py out/run_results.db
"""
import glob
import hashlib
import os
import sqlite3
import sys
from collections import defaultdict


def source_hashes():
    """'file=sha256[:16]' of every kernel source of the tree this script lives in: bench.py compares them with the tree it
    runs from and flags a roofline line whose instruction counts were profiled on other code (roofline.stale)."""
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tuch_amd', 'csrc')
    out = []
    for f in sorted(glob.glob(os.path.join(csrc, '*'))):
        if f.endswith(('.hip', '.h')):
            out.append('%s=%s' % (os.path.basename(f), hashlib.sha256(open(f, 'rb').read()).hexdigest()[:16]))
    return ' '.join(out)


def main(path):
    c = sqlite3.connect(path)
    cols = [d[1] for d in c.execute("pragma table_info('counters_collection')")]
    print('# rocprofv3 --pmc summary of %s' % path)
    print('# source sha256: %s' % source_hashes())
    name_col = 'kernel_name' if 'kernel_name' in cols else 'name'
    q = "select %s, counter_name, value, dispatch_id from counters_collection" % name_col
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for kname, cname, value, did in c.execute(q):
        acc[kname][cname] += float(value)
        disp[kname].add(did)
    print('%-60s %8s  %s' % ('kernel', 'launches', 'counter = average per launch'))
    for kname in sorted(acc, key=lambda k: -sum(acc[k].values())):
        n = max(len(disp[kname]), 1)
        vals = '  '.join('%s=%.6g' % (cn, acc[kname][cn] / n) for cn in sorted(acc[kname]))
        print('%-60s %8d  %s' % (kname[:60], n, vals))


if __name__ == '__main__':
    main(sys.argv[1])
