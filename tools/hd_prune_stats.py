import sys, types; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import bench
from tuch_amd.train.loss import RegressorLoss
dev = torch.device('cuda:0')
p = bench.build_problem(64, dev, 1002)
body = p['body']
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices
valid = torch.ones(64, dtype=torch.bool, device=dev)
crit = RegressorLoss(types.SimpleNamespace(contact_loss_weight=1.0), dev, body.num_verts, p['face_tensor'],
                     torch.tensor(body.geodesics, device=dev), geothres=0.3, euclthres=0.02,
                     face_tensor=p['face_tensor'], use_hd=True, segments=p['segments'],
                     hd_regressor=(body.hd_bary_idx, body.hd_bary_w), hd_faces=body.hd_face_id)
cap = {}
orig = crit._model.v2v_min_indexed
def spy(pts, vid, off, nmax, tree_order=False):
    cap['a'] = (pts.detach().cpu().numpy(), vid.cpu().numpy(), off.cpu().numpy(), nmax)
    return orig(pts, vid, off, nmax, tree_order=tree_order)
crit._model.v2v_min_indexed = spy
crit.contact_loss(verts, valid)
pts, vid, off, nmax = cap['a']
gm = body.geodesics > 0.3
Q = crit._model.tree_positions(); inv = np.argsort(Q); GM = gm[np.ix_(inv, inv)]
counts = np.diff(off)
print('points per body: mean %.0f max %d min %d, sum n^2 %.3g' % (counts.mean(), counts.max(), counts.min(), (counts.astype(np.float64) ** 2).sum()))
b = int(np.argmax(counts)); lo = off[b]; n = counts[b]
P = pts[lo:lo + n].astype(np.float64); vv = vid[lo:lo + n]
runs = 1 + (np.diff(vv) != 0).sum(); print('body', b, 'n', n, 'runs of equal vid', runs)
D = ((P[:, None] - P[None]) ** 2).sum(2)
allowed = GM[vv[None, :], vv[:, None]]
Dm = np.where(allowed, D, np.inf)
best = Dm.min(1); print('final best dist: median %.4f p90 %.4f max %.4f; inf %d' % (np.sqrt(np.median(best[np.isfinite(best)])), np.sqrt(np.percentile(best[np.isfinite(best)], 90)), np.sqrt(best[np.isfinite(best)].max()), (~np.isfinite(best)).sum()))
ub = Dm[:, ::8].min(1)
print('pass-1 bound: median %.4f p90 %.4f' % (np.sqrt(np.median(ub[np.isfinite(ub)])), np.sqrt(np.percentile(ub[np.isfinite(ub)], 90))), 'inf', (~np.isfinite(ub)).sum())
nch = (n + 31) // 32
cl = np.stack([P[c * 32:(c + 1) * 32].min(0) for c in range(nch)]); ch = np.stack([P[c * 32:(c + 1) * 32].max(0) for c in range(nch)])
print('chunk diag median %.3f' % np.median(np.linalg.norm(ch - cl, axis=1)))
proc = 0; tot = 0
for blk in range(0, n, 64):
    cols = slice(blk, min(n, blk + 64))
    e = np.maximum(np.maximum(cl[None] - P[cols][:, None], P[cols][:, None] - ch[None]), 0)
    lb = (e ** 2).sum(2)
    keep = (lb <= ub[cols][:, None]).any(0)
    proc += keep.sum(); tot += nch
print('chunks processed %.3f (with pass-1 bound, static)' % (proc / tot))
