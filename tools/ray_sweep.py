import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from tuch_amd.smplify.losses import contact_model_for
dev = torch.device('cuda:0')
for B in (64, 8):
    p = bench.build_problem(B, dev, 1002)
    model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
    with torch.no_grad():
        verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices
    for c in (1, 2, 4, 8, 12, 16):
        os.environ['TUCH_RAY_CHUNKS'] = str(c)
        t = bench.time_kernel(lambda: model.exterior_flags(verts, apply_segments=False), 20)
        print('B=%d chunks=%d: exterior_flags %.1f us' % (B, c, t * 1e6))
