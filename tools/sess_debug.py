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
gt = t(s['gt'])
def fit(f, use_contact=True):
    return f(pose, t(s['be']), t(s['cam_t']), torch.zeros(batch, 2, device=DEV), kp, use_contact=use_contact, contactlist=s['cdict'],
             gt_contact=[gt, None], ignore_idxs=torch.tensor([False, True, False], device=DEV),
             has_discrete_contact=torch.ones(batch, dtype=torch.bool, device=DEV), contact_loss_weight=2000.0, segments=s['segments'])
mk = lambda: SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=8, focal_length=5000., geodistssmpl=t(body.geodesics), geothres=0.3,
                       euclthres=0.02, device=torch.device(DEV), smpl=s['smpl'], pose_prior=s['prior'])
for uc in (True, False):
    f = mk()
    r1 = [x.clone() for x in fit(f, uc)[:6]]
    r2 = [x.clone() for x in fit(f, uc)[:6]]
    r3 = [x.clone() for x in fit(f, uc)[:6]]
    os.environ['TUCH_SMPLIFY_SESSIONS'] = '0'
    r0 = [x.clone() for x in fit(mk(), uc)[:6]]
    os.environ['TUCH_SMPLIFY_SESSIONS'] = '1'
    names = ('verts', 'joints', 'pose', 'betas', 'cam', 'reproj')
    for n, a, b, c, d in zip(names, r1, r2, r3, r0):
        print('use_contact', uc, n, 'call1-vs-fresh %.2e' % float((a - d).abs().max()), 'call2-vs-call1 %.2e' % float((b - a).abs().max()),
              'call3-vs-call2 %.2e' % float((c - b).abs().max()))
print('---- per-iteration losses')
Stage = SMPLifyDC._Stage
orig_run = Stage.run
def run(self, num_iters, collect):
    self._losses = []
    orig_one = self._one
    for state in self.optimizer.state.values():
        print(self.name, 'adam state before reset', {k: (float(v.abs().max()) if torch.is_tensor(v) else v) for k, v in state.items()})
    r = orig_run(self, num_iters, collect)
    return r
Stage.run = run
f = mk()
for call in range(3):
    fit(f, False)
    torch.cuda.synchronize()
    sess = list(f._sessions.values())[0]
    print('call', call, 'stage1 last loss', float(sess['stage1'].loss), 'stage2 last loss', float(sess['stage2_obj'].loss),
          'betas', sess['t']['betas'][0, :3].tolist())
