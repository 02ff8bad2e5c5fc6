import sys, os, faulthandler; faulthandler.enable(); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, numpy as np
import tests.test_gpu_smplify as T
DEV='cuda:0'
var=sys.argv[1]
batch=4
s=T._setup(batch,5); t=s['t']
be=t(s['be'])
if 'slices' in var:
    init_pose=torch.cat([t(s['go']),t(s['bp'])],1)
    bp=init_pose[:,3:].detach().clone(); go=init_pose[:,:3].detach().clone()
    bp.requires_grad=True; go.requires_grad=True
else:
    bp=t(s['bp']).requires_grad_(True); go=t(s['go']).requires_grad_(True)
opt=torch.optim.Adam([bp,go],lr=1e-2,capturable=True)
store={}
def one():
    out=s['smpl'](global_orient=go,body_pose=bp,betas=be)
    if 'jointsloss' in var: loss=(out.vertices**2).sum()+(out.joints**2).sum()
    else: loss=(out.vertices**2).sum()
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step(); store['v']=out.vertices
    return out.vertices
side=torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3 if 'warm3' in var else 2): one()
torch.cuda.current_stream().wait_stream(side)
if 'sync' in var: torch.cuda.synchronize()
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    one()
g.replay(); torch.cuda.synchronize()
print(var,'OK')
