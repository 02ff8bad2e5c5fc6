import os, sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
from helpers import golden
from oracle import contact as oc
from tuch_amd.ops import ContactModel
dev = torch.device('cuda:0')
for tag, batch in (('small', 3), ('medium', 9)):
    g = golden(tag)
    model = ContactModel(g['faces'], None, None, None, None, device=dev)
    base = torch.tensor(g['verts'], device=dev)
    verts = base[torch.arange(batch, device=dev) % base.shape[0]].contiguous()
    k = torch.arange(batch, device=dev, dtype=torch.float32).view(-1, 1)
    verts[:, :, 0] += 0.03 * k * verts[:, :, 1]
    verts[:, :, 2] += 0.1 * k
    res = {}
    for tree in (1, 0):
        os.environ['TUCH_WINDING_TREE'] = str(tree)
        res[tree] = model.exterior_flags(verts, apply_segments=False, return_details=True)[1].cpu().numpy()
    vn = verts.cpu().numpy()
    for b in range(batch):
        ref = oc.winding_numbers(vn[b], vn[b][g['faces']])
        print(tag, b, 'tree-ref %.2e flat-ref %.2e tree-flat %.2e' % (np.abs(res[1][b] - ref).max(), np.abs(res[0][b] - ref).max(), np.abs(res[1][b] - res[0][b]).max()))
