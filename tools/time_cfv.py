import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import bench
from tuch_amd.smplify.losses import contact_model_for
dev = torch.device('cuda:0')
p = bench.build_problem(64, dev, 1002)
model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices
print('pairs', model.num_pairs)
for B in (64, 32, 8):
    v = verts[:B].contiguous()
    t = bench.time_kernel(lambda: model.region_pair_min(v, select=None, masked=False), 10)
    print('contact_from_verts B=%d: %.3f ms' % (B, t * 1e3))
