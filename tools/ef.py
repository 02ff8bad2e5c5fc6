import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tuch_amd.smplify.losses import contact_model_for
dev = torch.device('cuda:0')
p = bench.build_problem(64, dev, 1002)
model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices
for _ in range(10):
    model.exterior_flags(verts, apply_segments=False)
torch.cuda.synchronize()
if len(sys.argv) > 1:
    print(model.ray_work(verts))
