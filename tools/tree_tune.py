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
ref = None
for leaf in [int(x) for x in os.environ.get('LEAVES', '32,48,64,96,128').split(',')]:
    os.environ['TUCH_TREE_LEAF_FACES'] = str(leaf)
    model = ops.ContactModel(faces, device=dev)
    for waves in [int(x) for x in os.environ.get('WAVES', '8192,32768,131072').split(',')]:
        model.set_option('tree_waves', waves)
        for B in (64, 8, 1):
            v = verts[:B].contiguous()
            t = bench.time_kernel(lambda: model.exterior_flags(v, apply_segments=False), 10)
            print('leaf %3d waves %6d B %2d: %.3f ms' % (leaf, waves, B, t * 1e3), flush=True)
    ext, det = model.exterior_flags(verts, apply_segments=False, return_details=True)[:2] if False else (None, None)
model.set_option('winding_tree', 0)
for B in (64, 8, 1):
    v = verts[:B].contiguous()
    t = bench.time_kernel(lambda: model.exterior_flags(v, apply_segments=False), 10)
    print('flat B %2d: %.3f ms' % (B, t * 1e3))
