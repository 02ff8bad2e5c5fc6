"""Classes of the (64-column block, 32-row tile) pairs of the HD search: of the pairs that survive the box test with
perfect bounds, how many have NO admissible (column, row) pair, how many have ALL pairs admissible, how many are mixed --
in the model's point order (leaf of the face) and with the points ordered by the tree position of their mask vertex."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = bench.build_problem(B, dev, 1002)
crit = bench.regressor_loss(p, True)
from tuch_amd.utils.geometry import batch_rodrigues
full_pose = torch.cat([p['global_orient'], p['body_pose']], dim=1)
rot = batch_rodrigues(full_pose.reshape(-1, 3)).view(B, 24, 3, 3)
verts = p['smpl'](betas=p['betas'], body_pose=rot[:, 1:], global_orient=rot[:, :1], pose2rot=False).vertices.detach()
valid = torch.ones(B, dtype=torch.bool, device=dev)
crit.contact_loss(verts.requires_grad_(True), valid)
hm = crit._hd
counts, sel = hm.selection(hm.last_saved, B)
idx, w, face = hm._host
model = crit._model
pos = torch.tensor(model.tree_positions(), device=dev).long()
faces = torch.tensor(model.faces_np, device=dev).long()
geomask = p['geomask']


def classes(pts, tv, best, tag):
    n = pts.shape[0]
    adm = geomask[tv][:, tv]                                  # [row, col]
    nch, ncb = (n + 31) // 32, (n + 63) // 64
    pad, padc = nch * 32 - n, ncb * 64 - n
    P = torch.cat([pts, pts[-1:].expand(pad, 3)]) if pad else pts
    lo, hi = P.view(nch, 32, 3).min(1).values, P.view(nch, 32, 3).max(1).values
    e = torch.clamp(torch.maximum(lo[:, None] - pts[None], pts[None] - hi[:, None]), min=0)
    lb = (e ** 2).sum(-1)
    surv_col = lb <= best[None]
    sc = torch.cat([surv_col, torch.zeros(nch, padc, dtype=torch.bool, device=dev)], 1).view(nch, ncb, 64).any(-1)
    # padded rows / columns copy the last real one (so that they change no class)
    ri = torch.cat([torch.arange(n, device=dev), torch.full((pad,), n - 1, device=dev)])
    ci = torch.cat([torch.arange(n, device=dev), torch.full((padc,), n - 1, device=dev)])
    A = adm[ri][:, ci].view(nch, 32, ncb, 64)
    anyp = A.any(3).any(1)
    allp = A.all(3).all(1)
    s = sc.float().sum().item()
    none_ = (sc & ~anyp).float().sum().item() / s
    all_ = (sc & allp).float().sum().item() / s
    # distinct mask vertices per tile and per column block
    tvp = torch.cat([tv, tv[-1:].expand(pad)]).view(nch, 32)
    dr = np.mean([len(torch.unique(r)) for r in tvp])
    tvc = torch.cat([tv, tv[-1:].expand(padc)]).view(ncb, 64)
    dc = np.mean([len(torch.unique(r)) for r in tvc])
    print(f'  {tag}: n={n} tiles surviving the box test {sc.float().mean().item():.3f} of all; of those: none {none_:.3f} '
          f'all {all_:.3f} mixed {1 - none_ - all_:.3f}; distinct mask vertices per 32-row tile {dr:.1f}, per 64 columns {dc:.1f}')


for b in range(B):
    n = int(counts[b]); ids = torch.tensor(sel[b, :n], device=dev).long()
    I = torch.tensor(idx, device=dev).long()[ids]; W = torch.tensor(w, device=dev)[ids]
    pts = (verts[b].detach()[I] * W[..., None]).sum(1)
    fc = torch.tensor(face, device=dev).long()[ids]
    tv = faces[fc, 0]
    adm = geomask[tv][:, tv]
    d2 = ((pts[:, None] - pts[None]) ** 2).sum(-1)
    best = torch.where(adm, d2, torch.full_like(d2, float('inf'))).min(0).values
    print(f'body {b}: median partner distance {best[best < 1e9].sqrt().median().item():.3f} m')
    classes(pts, tv, best, 'model order   ')
    o = torch.argsort(pos[tv] * 100000 + fc, stable=True)
    classes(pts[o], tv[o], best[o], 'mask-vertex order')
