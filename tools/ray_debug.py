import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import golden_io as gio
from helpers import golden
from tuch_amd.ops import ContactModel
dev = torch.device('cuda:0')
tag, batch, seed = 'full', 7, 5
g = golden(tag)
base = torch.tensor(g['verts'], device=dev)
verts = base[torch.arange(batch, device=dev) % base.shape[0]].clone()
rng = np.random.default_rng(seed)
for b in range(base.shape[0], batch):
    a = torch.tensor(np.eye(3) + 0.25 * rng.standard_normal((3, 3)), dtype=torch.float32, device=dev)
    verts[b] = verts[b] @ a.T + torch.tensor(rng.standard_normal(3), dtype=torch.float32, device=dev)
    print('body', b, 'det', float(torch.det(a)))
verts = verts.contiguous()
model = ContactModel(g['faces'], None, None, None, None, device=dev)
os.environ['TUCH_WINDING_RAY'] = '0'
w_s = model.exterior_flags(verts, apply_segments=False, return_details=True)[1].cpu().numpy()
for waves in ('32768', '1', '1000000'):
    os.environ['TUCH_RAY_WAVES'] = waves
    os.environ['TUCH_WINDING_RAY'] = '2'
    w_r = model.exterior_flags(verts, apply_segments=False, return_details=True)[1].cpu().numpy()
    bad = np.argwhere(np.abs(w_r - w_s) > 0.5)
    print('waves', waves, 'bad', len(bad), bad[:10].tolist(), [(float(w_r[b, v]), float(w_s[b, v])) for b, v in bad[:10]])
# single-body runs of the offending bodies
for b in sorted(set(int(x[0]) for x in bad)):
    w1 = model.exterior_flags(verts[b:b + 1].contiguous(), apply_segments=False, return_details=True)[1].cpu().numpy()
    print('body', b, 'alone: bad', int((np.abs(w1[0] - w_s[b]) > 0.5).sum()))
