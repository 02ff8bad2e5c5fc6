import os, sys; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch, bench, numpy as np
from tuch_amd.smplify.losses import contact_model_for
dev = torch.device('cuda:0')
p = bench.build_problem(64, dev, 1002)
model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices
ext = model.exterior_flags(verts, apply_segments=False).cpu().numpy()
segs = p['segments']
tot = 0
for name in segs.names:
    vid = segs.segmentation[name].segment_vidx
    n = (ext[:, vid] == 0).sum(1)
    nf = segs.segmentation[name].segment_faces.shape[0]
    print(name, 'faces', nf, 'verts', len(vid), 'interior per body: mean %.1f max %d, bodies with any %d' % (n.mean(), n.max(), (n > 0).sum()))
    tot += n.sum()
print('total interior segment queries', tot)
# the interior counts in classes (segment_one_kernel deals them in groups of 64 lanes)
allc = []
for name in segs.names:
    vid = segs.segmentation[name].segment_vidx
    allc.append((ext[:, vid] == 0).sum(1))
allc = np.concatenate(allc)
edges = [0, 1, 9, 17, 33, 65, 129, 257, 100000]
for lo, hi in zip(edges[:-1], edges[1:]):
    sel = (allc >= lo) & (allc < hi)
    print('pairs with %d..%d interior vertices: %d (their vertices: %d, lanes issued in groups of 64: %d)' % (lo, hi - 1, sel.sum(), allc[sel].sum(), (((allc[sel] + 63) // 64) * 64).sum()))
