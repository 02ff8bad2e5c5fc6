import sys, time; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch
import bench
dev=torch.device('cuda:0')
p=bench.build_problem(64, dev, 1002)
from tuch_amd.smplify.losses import contact_model_for
model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices
t=bench.time_kernel
a=t(lambda: model.exterior_flags(verts, apply_segments=False),10)
b=t(lambda: model.exterior_flags(verts, apply_segments=True),10)
c=t(lambda: model.exterior_flags(verts, apply_segments=True, return_details=True),10)
ext=model.exterior_flags(verts, apply_segments=False)
print('no seg %.3f ms, seg(skip) %.3f ms, seg(full) %.3f ms'%(a*1e3,b*1e3,c*1e3), 'interior frac', 1-ext.float().mean().item())
# how many 128-query groups of each segment contain an interior vertex
e=ext.cpu().numpy()
for (vidx,bands),name in zip(p['segments'].tables(), p['segments'].names):
    n=0;tot=0
    for bb in range(64):
        for q in range(0,len(vidx),64):
            tot+=1; n+= (e[bb][vidx[q:q+64]]==0).any()
    print(name,len(vidx),'waves needing work %d/%d'%(n,tot))
