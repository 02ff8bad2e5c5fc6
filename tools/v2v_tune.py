import os, sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import bench
dev = torch.device('cuda:0')
p = bench.build_problem(64, dev, 1002)
from tuch_amd.smplify.losses import contact_model_for
model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices
model.set_option('v2v_tree', 0)
mn0, a0 = model.v2v_min(verts)
for B in (64, 8, 1):
    v = verts[:B].contiguous()
    print('flat B %2d: %.3f ms' % (B, bench.time_kernel(lambda: model.v2v_min(v), 10) * 1e3))
model.set_option('v2v_tree', 1)
for waves in [int(x) for x in os.environ.get('WAVES', '4096,8192,16384,32768,65536').split(',')]:
    model.set_option('v2v_waves', waves)
    mn1, a1 = model.v2v_min(verts)
    torch.cuda.synchronize()
    same = (a0 == a1).float().mean().item()
    print('waves %6d: min equal %s  max|diff| %.3g  argmin equal %.5f' % (waves, torch.equal(mn0, mn1), (mn0 - mn1).abs().nan_to_num(0).max().item(), same))
    for B in (64, 8, 1):
        v = verts[:B].contiguous()
        print('   tree B %2d: %.3f ms' % (B, bench.time_kernel(lambda: model.v2v_min(v), 10) * 1e3), flush=True)
