import sys, faulthandler; faulthandler.enable(); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, numpy as np
import tests.test_gpu_smplify as T
from tuch_amd.smplify import smplifydc as M
from tuch_amd.utils.geometry import perspective_projection
DEV='cuda:0'
mode = sys.argv[1]
batch=4
s=T._setup(batch,5); body,t=s['body'],s['t']
fitter=M.SMPLifyDC(step_size=1e-2,batch_size=batch,num_iters=6,focal_length=5000.,geodistssmpl=t(body.geodesics),geothres=0.3,euclthres=0.02,device=torch.device(DEV),smpl=s['smpl'],pose_prior=s['prior'])
orig=fitter._optimise
calls={'n':0}
def wrapped(params, iteration, num_iters, adam_kwargs, collect=None):
    calls['n']+=1
    if mode=='stage2only' and calls['n']==1: fitter.use_graph=False
    else: fitter.use_graph=True
    if mode=='stage1only' and calls['n']==2: fitter.use_graph=False
    print('optimise call',calls['n'],'graph',fitter.use_graph, flush=True)
    if os.environ.get('SKIP1') and calls['n']==1:
        print(' skipped'); return
    if os.environ.get('NOCOLLECT'): collect=None
    r=orig(params, iteration, num_iters, adam_kwargs, collect)
    torch.cuda.synchronize(); print(' done', flush=True)
    return r
import os
fitter._optimise=wrapped
variant=os.environ.get('LOSSVAR','real')
_real=M.contact_fitting_loss
def patched(*a, **k):
    if variant=='v0': return (k['verts']**2).sum() + (a[5]**2).sum()
    if variant=='noseg': k['segments']=None
    if variant=='nor2r': k['gt_contact']=None
    if variant=='noprior':
        a=list(a); a[12]=(lambda pose, betas: torch.zeros(pose.shape[0], device=pose.device)); a=tuple(a)
    return _real(*a, **k)
M.contact_fitting_loss=patched

_en, _ex = torch.cuda.graph.__enter__, torch.cuda.graph.__exit__
def en(self):
    print('  capture enter...', flush=True); r=_en(self); print('  capture begun', flush=True); return r
def ex(self,*a):
    print('  capture ending...', flush=True); r=_ex(self,*a); print('  capture ended', flush=True); return r
torch.cuda.graph.__enter__=en; torch.cuda.graph.__exit__=ex
import tuch_amd.ops as O
for name in ['exterior_flags','v2v_min','region_pair_min']:
    f=getattr(O.ContactModel,name)
    def mk(f,name):
        def w(self,*a,**k):
            print('   ',name,flush=True); return f(self,*a,**k)
        return w
    setattr(O.ContactModel,name,mk(f,name))

with torch.no_grad():
    tgt=s['smpl'](global_orient=t(s['go']),body_pose=t(s['bp'])+0.1,betas=t(s['be']))
    j2d=perspective_projection(tgt.joints,torch.eye(3,device=DEV)[None].expand(batch,-1,-1),t(s['cam_t']),5000.,torch.zeros(batch,2,device=DEV))
kp=torch.cat([j2d,torch.ones(batch,49,1,device=DEV)],2)
init_pose=torch.cat([t(s['go']),t(s['bp'])],1)
res=fitter(init_pose,t(s['be']),t(s['cam_t']),torch.zeros(batch,2,device=DEV),kp.clone(),use_contact=True,contactlist=s['cdict'],gt_contact=[t(s['gt']),None],ignore_idxs=torch.zeros(batch,dtype=torch.bool,device=DEV),has_discrete_contact=torch.ones(batch,dtype=torch.bool,device=DEV),contact_loss_weight=1.0,segments=s['segments'])
print('OK', res[0].shape)
