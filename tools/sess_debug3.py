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
for zero in ('', 'again'):
    f = mk()
    rs = []
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    for c in range(6):
        if zero == 'side':
            with torch.cuda.stream(st):
                rs.append([x.clone() for x in fit(f, False)[:6]])
        else:
            rs.append([x.clone() for x in fit(f, False)[:6]])
        if zero == 'syncafter':
            torch.cuda.synchronize()
    # inspect adam state right after a call, then what a call sees
    print('zero mode %r: pose diffs vs call1:' % zero, ['%.1e' % float((r[2] - rs[0][2]).abs().max()) for r in rs[1:]],
          'cam diffs', ['%.1e' % float((r[4] - rs[0][4]).abs().max()) for r in rs[1:]])
# which stage?  stage-1 only quantities: cam (stage 1 optimises cam + global_orient without contact)
