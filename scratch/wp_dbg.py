import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
from helpers import golden
from oracle import contact as oc
from tuch_amd.ops import ContactModel
dev = torch.device('cuda:0')
g = golden('full')
model = ContactModel(g['faces'], None, None, None, None, device=dev)
verts_np = g['verts']; b_count, v_count = verts_np.shape[:2]
rng = np.random.default_rng(3); q = 300
pts = np.stack([verts_np[b][rng.choice(v_count, q, replace=False)] for b in range(b_count)])
pts = (pts + 0.003 * rng.standard_normal(pts.shape)).astype(np.float32); pts[:, :5] += 3.0
w, ext = model.winding_points(torch.tensor(verts_np, device=dev), torch.tensor(pts, device=dev), None)
w = w.cpu().numpy()
wo = oc.winding_numbers(pts[0], oc.gather_tris(verts_np[0], g['faces']))
err = np.abs(w[0] - wo); k = np.argsort(err)[-5:]
print('p99 %.3e max %.3e' % (np.percentile(err, 99), err.max()), err[k], wo[k], w[0][k])
# float64 truth
v = verts_np[0].astype(np.float64); tri = v[g['faces']]
for i in k:
    p = pts[0][i].astype(np.float64)
    a, b, c = tri[:, 0] - p, tri[:, 1] - p, tri[:, 2] - p
    la, lb, lc = [np.linalg.norm(x, axis=1) for x in (a, b, c)]
    num = np.einsum('tk,tk->t', a, np.cross(b, c)); den = la * lb * lc + (a * b).sum(1) * lc + (a * c).sum(1) * lb + (b * c).sum(1) * la
    print(i, 'f64 %.7f oracle %.7f gpu %.7f  min dist to vertex %.2e' % (np.arctan2(num, den).sum() / (2 * np.pi), wo[i], w[0][i], np.linalg.norm(v - p, axis=1).min()))
