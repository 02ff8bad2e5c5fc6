"""How many vertices scatter their gradient into the SAME partner vertex (the stage-2 tail's atomics on one address)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from tuch_amd.smplify.losses import contact_model_for
dev = torch.device('cuda:0'); torch.cuda.set_stream(torch.cuda.Stream(device=dev))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = bench.build_problem(B, dev, 1002)
model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices.contiguous()
ext = model.exterior_flags(verts).bool()
mn, partner = model.v2v_min(verts)
d = mn.sqrt()
active = (~ext) | (d < 0.02)
V = verts.shape[1]
worst = []
for b in range(B):
    pa = partner[b][active[b]].long()
    cnt = torch.bincount(pa, minlength=V)
    worst.append(int(cnt.max()))
    if b < 3:
        top = torch.topk(cnt, 5).values.tolist()
        print('body %d: %d active vertices, %d distinct partners, largest in-degrees %s' % (b, int(active[b].sum()), int((cnt > 0).sum()), top))
print('largest in-degree per body: max %d, mean %.1f' % (max(worst), sum(worst) / len(worst)))
# multiplicity inside wavefronts of 64 consecutive vertices
pa = torch.where(active, partner.long(), torch.full_like(partner.long(), -1))
pad = (-V) % 64
pa = torch.nn.functional.pad(pa, (0, pad), value=-1).view(B, -1, 64)
tot = dup = 0
for w in pa.view(-1, 64)[:4000]:
    x = w[w >= 0]
    tot += len(x); dup += len(x) - len(torch.unique(x))
print('inside wavefronts: %d active lanes, %d of them repeat a partner of the same wavefront' % (tot, dup))
