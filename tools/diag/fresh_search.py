"""The nearest-vertex search on FRESH bodies (a training loop never sees the same bodies twice) against the same search
with the previous call's partners as its seed (an iterative fit): the search alone, the inside test alone (no state between
calls: the control), and the train.py-style contact step, each with
  same     -- identical vertices every call (the hint is the previous call's exact answer)
  rotate   -- K distinct pose batches in turn (the hint is another batch's answer: a foreign hint)
  nohint   -- option v2v_hint = 0 (two launches, seed from the nearest admissible leaf)
  zeroed   -- the hint buffer cleared before every call (a first call)

    python tools/diag/fresh_search.py [batch] [K]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from tuch_amd.smplify.losses import contact_model_for

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
probs = [bench.build_problem(B, dev, 1002 + 17 * k) for k in range(K)]
p = probs[0]
model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
with torch.no_grad():
    verts = [q['smpl'](global_orient=q['global_orient'], body_pose=q['body_pose'], betas=q['betas']).vertices.clone()
             for q in probs]


def timed(fn, n=24):
    for _ in range(K + 1):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


class Rot:
    def __init__(self):
        self.i = 0

    def next(self):
        self.i = (self.i + 1) % K
        return verts[self.i]


rot = Rot()
res = {}
ref = [model.v2v_min(v) for v in verts]
res['search same'] = timed(lambda: model.v2v_min(verts[0]))
res['search rotate'] = timed(lambda: model.v2v_min(rot.next()))
# results never depend on the hints
for k in range(K):
    mn, arg = model.v2v_min(verts[k])
    assert torch.equal(mn, ref[k][0]) and torch.equal(arg, ref[k][1]), 'foreign hints changed the result'
hint = model._v2v_hint(B)


def zeroed():
    hint.zero_()
    return model.v2v_min(verts[0])


res['search zeroed (incl. the fill)'] = timed(zeroed)
model.set_option('v2v_hint', 0)
res['search nohint same'] = timed(lambda: model.v2v_min(verts[0]))
res['search nohint rotate'] = timed(lambda: model.v2v_min(rot.next()))
for k in range(K):
    mn, arg = model.v2v_min(verts[k])
    assert torch.equal(mn, ref[k][0]) and torch.equal(arg, ref[k][1]), 'no hints changed the result'
model.set_option('v2v_hint', 1)
res['inside same'] = timed(lambda: model.exterior_flags(verts[0]))
res['inside rotate'] = timed(lambda: model.exterior_flags(rot.next()))
for k, v in res.items():
    print('%-36s %8.1f us' % (k, v))

# the train.py-style contact step (SMPL forward with rotation matrices -> RegressorLoss.contact_loss -> backward), one graph,
# the pose batch written in place between replays
from tuch_amd.utils.geometry import batch_rodrigues
for use_hd in (False, True):
    crit = bench.regressor_loss(p, use_hd)
    rots = [batch_rodrigues(torch.cat([q['global_orient'], q['body_pose']], 1).reshape(-1, 3)).view(B, 24, 3, 3) for q in probs]
    bets = [q['betas'].clone() for q in probs]
    rotmat = rots[0].clone().requires_grad_(True)
    betas = bets[0].clone().requires_grad_(True)
    valid = torch.ones(B, dtype=torch.bool, device=dev)

    def step():
        rotmat.grad = betas.grad = None
        o = p['smpl'](betas=betas, body_pose=rotmat[:, 1:], global_orient=rotmat[:, :1], pose2rot=False)
        loss = crit.contact_loss(o.vertices, valid)
        loss.backward()
        return loss
    replay = bench.capture(step, 3)
    state = {'i': 0}

    def fresh():
        state['i'] = (state['i'] + 1) % K
        with torch.no_grad():
            rotmat.copy_(rots[state['i']])
            betas.copy_(bets[state['i']])
        return replay()

    def same():
        with torch.no_grad():
            rotmat.copy_(rots[0])
            betas.copy_(bets[0])
        return replay()
    print('train-style %-5s same %8.1f us   fresh %8.1f us' % ('hd' if use_hd else 'plain', timed(same, 12), timed(fresh, 12)))
