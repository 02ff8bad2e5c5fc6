import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [d[1] for d in c.execute("pragma table_info('kernels')")]
rows = list(c.execute("select name, start, end, stream_id from kernels order by start"))
# find last two winding_tree launches of the model path (grid big): take steps near the end of the timed loop
idx = [i for i, r in enumerate(rows) if 'small_terms_kernel' in r[0]]
a, b = idx[-4], idx[-3]
t0 = rows[a][1]
prev_end = None
tot = 0
for r in rows[a:b]:
    gap = (r[1] - prev_end) / 1e3 if prev_end else 0
    dur = (r[2] - r[1]) / 1e3
    tot += dur
    print('%8.1f  +%6.1f gap  %7.1f us  s%s  %s' % ((r[1] - t0) / 1e3, gap, dur, r[3], r[0][:90]))
    prev_end = max(prev_end or 0, r[2])
print('step span %.1f us, kernel sum %.1f us, kernels %d' % ((rows[b][1] - t0) / 1e3, tot, b - a))
