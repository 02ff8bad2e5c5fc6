import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
import bench
from tuch_amd.smplify.smplifydc import SMPLifyDC
dev=torch.device('cuda:0')
B=int(sys.argv[1]); iters=int(sys.argv[2])
p=bench.build_problem(B, dev, 1003)
for use_graph in (True, False):
    fitter=SMPLifyDC(step_size=1e-2,batch_size=B,num_iters=iters,focal_length=5000.,geodistssmpl=torch.tensor(p['body'].geodesics,device=dev),geothres=0.3,euclthres=0.02,device=dev,smpl=p['smpl'],pose_prior=p['prior'],use_graph=use_graph)
    kp=torch.cat([p['j2d'],p['conf'][...,None]],2)
    init_pose=torch.cat([p['global_orient'],p['body_pose']],1)
    def run():
        return fitter(init_pose,p['betas'],p['cam_t'],p['cam_c'],kp.clone(),use_contact=True,contactlist=p['cdict'],gt_contact=[p['gt'],None],ignore_idxs=p['ignore'],has_discrete_contact=p['has_dc'],contact_loss_weight=2000.0,segments=p['segments'])
    run(); torch.cuda.synchronize()
    t0=time.time(); r=run(); torch.cuda.synchronize(); dt=time.time()-t0
    print('B=%d iters=%d+%d use_graph=%s: %.3f s total, %.2f ms per stage-2 iteration equivalent'%(B,iters,iters,use_graph,dt,dt/iters*1e3/2))
