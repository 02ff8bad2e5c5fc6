import sys, faulthandler; faulthandler.enable(); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, numpy as np
import tests.test_gpu_smplify as T
from tuch_amd import ops
from tuch_amd.smplify.losses import contact_model_for, contact_fitting_loss
DEV='cuda:0'
which=sys.argv[1]
batch=4
s=T._setup(batch,5); body,t=s['body'],s['t']
face_tensor=t(body.faces)[None].repeat(batch,1,1)
geomask=t(body.geodesics)>0.3
model=contact_model_for(geomask, face_tensor, s['segments'], s['cdict'])
be=t(s['be']); camt=t(s['cam_t'])
bp=t(s['bp']).requires_grad_(True); go=t(s['go']).requires_grad_(True)
verts=s['smpl'](global_orient=go,body_pose=bp,betas=be).vertices.detach()
vg=verts.clone().requires_grad_(True)
ext=model.exterior_flags(verts,True); mn,arg=model.v2v_min(verts)
sel=(t(s['gt'])==1)
kp2=t(s['kp'][:,:,:2]); conf=t(s['kp'][:,:,2]).contiguous()
gtc=t(s['gt']); zc=torch.zeros(batch,2,device=DEV); ign=torch.zeros(batch,dtype=torch.bool,device=DEV); hdc=torch.ones(batch,dtype=torch.bool,device=DEV)
fns={
 'exterior': lambda: model.exterior_flags(verts, True),
 'exterior_noseg': lambda: model.exterior_flags(verts, False),
 'v2v': lambda: model.v2v_min(verts),
 'terms': lambda: ops.contact_terms(vg,arg,ext,None,0,0.02)[0].sum().backward(),
 'region': lambda: model.region_pair_min(vg, select=sel, masked=True)[0].sum().backward(),
 'smpl': lambda: s['smpl'](global_orient=go,body_pose=bp,betas=be).vertices.sum().backward(),
 'small': lambda: ops.smplify_small_terms(s['smpl'](global_orient=go,body_pose=bp,betas=be).joints, camt, bp, zc, kp2, conf, s['prior'].means, s['prior'].precisions, s['prior'].log_nll_weights, 5000., 100., 1.0).sum().backward(),
}
def full():
    out=s['smpl'](global_orient=go,body_pose=bp,betas=be)
    loss=contact_fitting_loss(bp,go,None,None,be,out.joints,geomask,0.02,camt,zc,kp2,conf,s['prior'],cdict=s['cdict'],gt_contact=[gtc,None],ignore_idxs=ign,has_discrete_contact=hdc,verts=out.vertices,face_tensor=face_tensor,contact_loss_weight=1.0,segments=s['segments'])
    loss.backward()
fns['full']=full
variant = sys.argv[2] if len(sys.argv) > 2 else ''
if which.startswith('opt'):
    kw = dict(capturable=True)
    if 'fused' in variant: kw['fused'] = True
    opt = torch.optim.Adam([bp, go], lr=1e-2, **kw)
    store = {}
    def optstep():
        out=s['smpl'](global_orient=go,body_pose=bp,betas=be)
        loss=contact_fitting_loss(bp,go,None,None,be,out.joints,geomask,0.02,camt,zc,kp2,conf,s['prior'],cdict=s['cdict'],gt_contact=[gtc,None],ignore_idxs=ign,has_discrete_contact=hdc,verts=out.vertices,face_tensor=face_tensor,contact_loss_weight=1.0,segments=s['segments'])
        opt.zero_grad(set_to_none=('keepgrad' not in variant))
        loss.backward()
        opt.step()
        store['v']=out.vertices
    fns[which]=optstep
fn=fns[which]
side=torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fn(); fn()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fn()
g.replay(); torch.cuda.synchronize()
print(which,'OK')
