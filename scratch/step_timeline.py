import sqlite3, sys
c=sqlite3.connect(sys.argv[1])
rows=list(c.execute("select name,start,end from kernels order by start"))
idx=[i for i,r in enumerate(rows) if "pose_kernel" in r[0]]
# choose a step in the middle of the timed region
a,b=idx[len(idx)//2], idx[len(idx)//2+1]
seg=rows[a:b]
t0=seg[0][1]
busy=0
for n,s,e in seg:
    busy+=e-s
    print('%9.1f %8.1f  %s'%((s-t0)/1e3,(e-s)/1e3,n[:90]))
print('kernels',len(seg),'busy %.3f ms span %.3f ms'%(busy/1e6,(seg[-1][2]-t0)/1e6))
