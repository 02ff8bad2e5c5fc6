import sys, time; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, torch
from synthetic import make_body, random_poses
from tuch_amd import ops
from oracle import lbs as ol
dev=torch.device('cuda:0')
B=int(sys.argv[1]) if len(sys.argv)>1 else 64
t=time.time(); body=make_body(84,82); print('model %.1fs'%(time.time()-t))
m=ol.model_tensors(body)
bp,go,be=random_poses(B,1002)
v,_=ol.smpl_forward(m,torch.tensor(be),torch.tensor(bp),torch.tensor(go))
verts=v.to(dev).contiguous()
names=list(body.regions.keys())
pairs=np.asarray([[names.index(a),names.index(b)] for a,b in body.region_pairs])
model=ops.ContactModel(body.faces, body.geodesics>0.3, [(s['vidx'],list(s['bands'].values())) for s in body.segments.values()], [body.regions[n] for n in names], pairs, device=dev)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
V,F=body.num_verts, body.num_faces
t_ext=timeit(lambda: model.exterior_flags(verts, apply_segments=False))
t_seg=timeit(lambda: model.exterior_flags(verts, apply_segments=True))
t_v2v=timeit(lambda: model.v2v_min(verts))
ext=model.exterior_flags(verts); mn,arg=model.v2v_min(verts)
vg=verts.clone().requires_grad_(True)
def terms():
    pb,_=ops.contact_terms(vg,arg,ext,None,0,0.02); pb.sum().backward()
t_terms=timeit(terms)
t_reg=timeit(lambda: model.region_pair_min(verts))
pairs_w = B*V*F
print('B=%d winding %.3f ms  (%.1f TFLOP/s @67 flop/pair, %.1f%% of 157.3) | +segments %.3f ms | v2v %.3f ms | terms fwd+bwd %.3f ms | region all-pairs %.3f ms'%(B,t_ext,pairs_w*67/t_ext/1e9,pairs_w*67/t_ext/1e9/157.3*100,t_seg,t_v2v,t_terms,t_reg))
print('interior frac', 1-ext.float().mean().item())
