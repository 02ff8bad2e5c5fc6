import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from tuch_amd import ops
dev = torch.device('cuda:0')
B = 64
p = bench.build_problem(B, dev, 1002)
crit = bench.regressor_loss(p, True)
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices
# monkeypatch to keep the workspace
orig = ops._workspace
keep = {}
def ws(nbytes, device):
    t = orig(nbytes, device); keep['last'] = t; return t
ops._workspace = ws
valid = torch.ones(B, dtype=torch.bool, device=dev)
with torch.no_grad():
    crit.contact_loss(verts, valid)
torch.cuda.synchronize()
N = crit._hd.num_points
al = lambda x: (x + 255) & ~255
w = keep['last']
o_offs = 0; o_vid = al(B * N * 12); o_sb = o_vid + al(B * N * 4); o_sa = o_sb + al(B * N * 4); o_md = o_sa + al(B * N * 4)
sb = w[o_sb:o_sb + B * N * 4].view(torch.float32).view(B, N)
md = w[o_md:o_md + B * N * 4].view(torch.float32).view(B, N)
counts, sel = crit._hd.selection(crit._hd.last_saved, B)
tot = fin = 0; ratios = []; blocks_all = blocks_tot = 0
for b in range(B):
    n = counts[b]
    s = sb[b, :n].cpu().numpy(); m = md[b, :n].cpu().numpy()
    tot += n; fin += np.isfinite(s).sum()
    ok = np.isfinite(s) & np.isfinite(m) & (m > 0)
    ratios.append(np.sqrt(s[ok] / m[ok]))
    for c in range(0, n, 64):
        blocks_tot += 1; blocks_all += bool(np.isfinite(s[c:c + 64]).all())
r = np.concatenate(ratios)
print('selected per body mean %.0f; seeds finite %.3f; blocks with all seeds %.3f; seed dist / final dist: median %.2f p90 %.2f p99 %.2f'
      % (tot / B, fin / tot, blocks_all / blocks_tot, np.median(r), np.percentile(r, 90), np.percentile(r, 99)))
print('final partner distance (m): median %.4f p90 %.4f' % (np.sqrt(np.median(np.concatenate([md[b, :counts[b]].cpu().numpy() for b in range(B)]))), 0))
