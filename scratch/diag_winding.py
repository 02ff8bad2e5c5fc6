import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch
from helpers import *
from oracle import contact as oc
from tuch_amd import ops
dev=torch.device('cuda:0')
for tag in ['small','medium','full']:
    g=golden(tag)
    verts=torch.tensor(g['verts'],device=dev); faces=torch.tensor(g['faces'].astype(np.int32),device=dev)
    tris=ops.gather_triangles(verts,faces)
    w=ops.winding_numbers(verts,tris).cpu().numpy()
    for b in range(w.shape[0]):
        wo = oc.winding_numbers(g['verts'][b], oc.gather_tris(g['verts'][b], g['faces']))
        # fp64 truth
        v=g['verts'][b].astype(np.float64); T=v[g['faces']]
        e=np.abs(w[b]-g['winding'][b]); eo=np.abs(w[b]-wo); er=np.abs(wo-g['winding'][b])
        print(tag,b,'gpu-ref: max %.2e p99.9 %.2e p99 %.2e med %.2e | gpu-oracle max %.2e | oracle-ref max %.2e p99.9 %.2e'%(e.max(),np.percentile(e,99.9),np.percentile(e,99),np.median(e),eo.max(),er.max(),np.percentile(er,99.9)), 'n>1e-5:',(e>1e-5).sum(),'of',len(e), 'flagdiff', ((w[b]<=0.99)!=(g['winding'][b]<=0.99)).sum())
