"""A few calls of the nearest-vertex search alone (iterative form, batch 64): the workload of tools/diag/search_pmc.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from tuch_amd.smplify.losses import contact_model_for
dev = torch.device('cuda:0')
p = bench.build_problem(64, dev, 1002)
model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices.clone()
for _ in range(6):
    model.v2v_min(verts, iterative=True)
torch.cuda.synchronize()
