import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
os.environ['TUCH_GRAPH_STRICT'] = '1'
import numpy as np, torch
import test_gpu_smplify as T
from tuch_amd.smplify.smplifydc import SMPLifyDC
DEV = 'cuda:0'
batch = 3
s = T._setup(batch, 31)
body, t = s['body'], s['t']
kp = torch.cat([torch.randn(batch, 49, 2, device=DEV) * 30, torch.rand(batch, 49, 1, device=DEV)], 2)
pose = torch.cat([t(s['go']), t(s['bp'])], 1)
gt = t(s['gt']); be = t(s['be']); cam = t(s['cam_t']); cc = torch.zeros(batch, 2, device=DEV)
ign = torch.tensor([False, True, False], device=DEV); hdc = torch.ones(batch, dtype=torch.bool, device=DEV)
def fit(f, use_contact=True):
    return f(pose, be, cam, cc, kp, use_contact=use_contact, contactlist=s['cdict'], gt_contact=[gt, None], ignore_idxs=ign,
             has_discrete_contact=hdc, contact_loss_weight=2000.0, segments=s['segments'])
mk = lambda: SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=8, focal_length=5000., geodistssmpl=t(body.geodesics), geothres=0.3,
                       euclthres=0.02, device=torch.device(DEV), smpl=s['smpl'], pose_prior=s['prior'])
for mode in ('', 'k', 'c'):
    os.environ['TUCH_SESS_SYNC'] = mode
    f = mk()
    r1 = [x.clone() for x in fit(f, False)[:6]]
    r2 = [x.clone() for x in fit(f, False)[:6]]
    print('sync mode %r (inputs held by the caller): call2 vs call1 pose %.2e betas %.2e' % (mode, float((r2[2] - r1[2]).abs().max()), float((r2[3] - r1[3]).abs().max())))
