import sqlite3, sys
from collections import defaultdict
c=sqlite3.connect(sys.argv[1]); pat=sys.argv[2]
rows=list(c.execute("select dispatch_id, counter_name, value, duration from counters_collection where kernel_name like ? order by dispatch_id", ('%'+pat+'%',)))
d=defaultdict(dict)
for did,cn,v,dur in rows:
    d[did][cn]=v; d[did]['dur_us']=dur/1e3
ids=list(d)
for did in ids[:2]+ids[-3:]:
    print(did, {k:('%.4g'%v) for k,v in sorted(d[did].items())})
