import os, sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import bench
from tuch_amd import ops
dev = torch.device('cuda:0')
p = bench.build_problem(64, dev, 1002)
body = p['body']
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices
faces = torch.as_tensor(body.faces.astype(np.int64))
model = ops.ContactModel(faces, device=dev)
for waves in (4096, 16384, 32768, 65536, 131072):
    model.set_option('tree_waves', waves)
    w = model.winding_tree_work(verts)
    t = bench.time_kernel(lambda: model.exterior_flags(verts, apply_segments=False), 10)
    print(waves, w, 'per block: leaf %.0f cap %.0f' % (w['leaf_elements'] / w['query_blocks'], w['cap_elements'] / w['query_blocks']), '%.3f ms' % (t * 1e3))
