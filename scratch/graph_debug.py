import sys, traceback; sys.path.insert(0,'.')
import torch, numpy as np
import bench
dev=torch.device('cuda:0')
p=bench.build_problem(8, dev, 1002)
from tuch_amd.smplify.losses import contact_fitting_loss, contact_model_for
from tuch_amd import ops
model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
bp = p['body_pose'].clone().requires_grad_(True); go = p['global_orient'].clone().requires_grad_(True)
verts = p['smpl'](global_orient=go, body_pose=bp, betas=p['betas']).vertices.detach()
valid = torch.ones(8, dtype=torch.uint8, device=dev)
def try_capture(name, fn):
    try:
        s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn(); fn()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g=torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay(); torch.cuda.synchronize()
        print(name,'OK')
    except Exception as e:
        print(name,'FAILED', type(e).__name__, str(e)[:120])
        torch.cuda.synchronize()
try_capture('smpl fwd', lambda: p['smpl'](global_orient=go, body_pose=bp, betas=p['betas']))
try_capture('smpl fwd+bwd', lambda: p['smpl'](global_orient=go, body_pose=bp, betas=p['betas']).vertices.sum().backward())
try_capture('exterior', lambda: model.exterior_flags(verts, True))
try_capture('v2v', lambda: model.v2v_min(verts))
ext=model.exterior_flags(verts, True); mn,arg=model.v2v_min(verts)
vg = verts.clone().requires_grad_(True)
try_capture('terms', lambda: ops.contact_terms(vg,arg,ext,valid,0,0.02)[0].sum().backward())
sel = (p['gt']==1)
try_capture('region', lambda: model.region_pair_min(vg, select=sel, masked=True)[0].sum().backward())
try_capture('prior', lambda: p['prior'](bp, p['betas']).sum().backward())
step, red = bench.make_step(p, 1)
try_capture('full step', step)
