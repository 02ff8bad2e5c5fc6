"""Where do the rows of the HD search go?  One HD forward at batch 8, then a simulation of v2v_indexed_kernel's pruning
in torch: per (64-column block, 32-row chunk) -- survives the box test / has any admissible pair / both."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
dev = torch.device('cuda:0')
B = 8
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
gm_tree = None
for b in range(B):
    n = int(counts[b]); ids = torch.tensor(sel[b, :n], device=dev).long()
    I = torch.tensor(idx, device=dev).long()[ids]; W = torch.tensor(w, device=dev)[ids]
    pts = (verts[b].detach()[I] * W[..., None]).sum(1)
    tv = faces[torch.tensor(face, device=dev).long()[ids], 0]
    adm = geomask[tv][:, tv]                                  # [n,n] admissible
    d2 = ((pts[:, None] - pts[None]) ** 2).sum(-1)
    d2m = torch.where(adm, d2, torch.full_like(d2, float('inf')))
    best = d2m.min(0).values                                  # per column
    nch, ncb = (n + 31) // 32, (n + 63) // 64
    pad = nch * 32 - n
    P = torch.cat([pts, pts[-1:].expand(pad, 3)]) if pad else pts
    lo, hi = P.view(nch, 32, 3).min(1).values, P.view(nch, 32, 3).max(1).values
    e = torch.clamp(torch.maximum(lo[:, None] - pts[None], pts[None] - hi[:, None]), min=0)   # [nch, n, 3]
    lb = (e ** 2).sum(-1)
    surv_col = lb <= best[None]                               # [nch, n]
    padc = ncb * 64 - n
    sc = torch.cat([surv_col, torch.zeros(nch, padc, dtype=torch.bool, device=dev)], 1).view(nch, ncb, 64).any(-1)   # [nch, ncb]
    admp = torch.cat([adm, torch.zeros(nch * 32 - n, n, dtype=torch.bool, device=dev)], 0)
    admp = torch.cat([admp, torch.zeros(nch * 32, padc, dtype=torch.bool, device=dev)], 1)
    ac = admp.view(nch, 32, ncb, 64).any(3).any(1)            # exact: chunk has a row admissible for some column
    blk = (pos[tv] >> 6)
    nb = int(blk.max()) + 1
    # block-level table
    posm = geomask[:, :]
    vb = (pos >> 6)
    T = torch.zeros(nb + 1, nb + 1, dtype=torch.bool, device=dev)
    vbk = vb.clamp(max=nb)
    # any admissible between vertex blocks
    onehot = torch.zeros(geomask.shape[0], nb + 1, device=dev); onehot[torch.arange(geomask.shape[0]), vbk] = 1
    T = (onehot.t() @ geomask.float() @ onehot) > 0
    rb = torch.cat([blk, blk[-1:].expand(nch * 32 - n)]).view(nch, 32)
    cb = torch.cat([blk, blk[-1:].expand(padc)]).view(ncb, 64)
    bt = torch.zeros(nch, ncb, dtype=torch.bool, device=dev)
    for i in range(32):
        for j in range(0, 64, 8):
            bt |= T[rb[:, i][:, None], cb[:, j][None]]
    print(f'body {b}: n={n} median best={best[best < 1e9].sqrt().median().item():.3f} m  box-survive={sc.float().mean().item():.3f} '
          f'exact-adm={ac.float().mean().item():.3f} box&exact={(sc & ac).float().mean().item():.3f} '
          f'block-adm~={bt.float().mean().item():.3f} box&block={(sc & bt).float().mean().item():.3f} '
          f'table density={T[:nb,:nb].float().mean().item():.3f} per-column survive={surv_col.float().mean().item():.3f}')
