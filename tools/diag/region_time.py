"""region_pair_min timing vs number of selected pairs."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from tuch_amd.smplify.losses import contact_model_for
dev = torch.device('cuda:0'); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = bench.build_problem(B, dev, 1002)
model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices.contiguous()
P = model.num_pairs
print('pairs', P, 'region sizes', sorted(len(v) for v in p['cdict']['csig'].values())[-5:])
for name, sel in (('none', torch.zeros(B, P, dtype=torch.uint8, device=dev)), ('bench', (p['gt'] == 1).to(torch.uint8)),
                  ('one per body', torch.nn.functional.one_hot(torch.zeros(B, dtype=torch.long), P).to(torch.uint8).to(dev)),
                  ('all', torch.ones(B, P, dtype=torch.uint8, device=dev))):
    for masked in (True, False):
        t = bench.time_kernel(lambda: model.region_pair_min(verts, select=sel.contiguous(), masked=masked), 10)
        print('%-14s masked=%d selected %6d: %.1f us' % (name, masked, int(sel.sum()), t * 1e6), flush=True)
