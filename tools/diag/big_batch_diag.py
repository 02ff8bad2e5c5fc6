"""Diagnose ray-vs-solid-angle flag mismatches of test_big_batches_* (prints w of both paths and the float64 value)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np, torch
import test_gpu_contact as T
tag, batch = sys.argv[1], int(sys.argv[2])
g, verts = T._posed_batch(tag, batch, 5)
model = T.make_model(g, T.golden_mask(tag), True, False)
model.set_option('winding_ray', 0)
e0, w0 = model.exterior_flags(verts, apply_segments=False, return_details=True)[:2]
model.set_option('winding_ray', 2)
e1, w1 = model.exterior_flags(verts, apply_segments=False, return_details=True)[:2]
model.set_option('winding_tree', 0); model.set_option('winding_ray', 0)
e2, w2 = model.exterior_flags(verts, apply_segments=False, return_details=True)[:2]
bad = torch.nonzero((e0 != e1) | ((w0 - w1).abs() > 0.5)).cpu().numpy()
faces = g['faces']
for b, vid in bad:
    v = verts[b].cpu().numpy().astype(np.float64)
    tri = v[faces[~(faces == vid).any(1)]]
    a, bb, c = tri[:, 0] - v[vid], tri[:, 1] - v[vid], tri[:, 2] - v[vid]
    la, lb, lc = [np.linalg.norm(x, axis=1) for x in (a, bb, c)]
    num = np.einsum('ij,ij->i', a, np.cross(bb, c))
    den = la * lb * lc + (a * bb).sum(1) * lc + (a * c).sum(1) * lb + (bb * c).sum(1) * la
    w64 = np.arctan2(num, den).sum() / (2 * np.pi)
    print('body %d vertex %d: w tree-solid %.6f  ray %.6f  flat-solid %.6f  float64 %.6f  valence %d' % (
        b, vid, w0[b, vid].item(), w1[b, vid].item(), w2[b, vid].item(), w64, int((faces == vid).any(1).sum())))
print('mismatches', len(bad))
# ---- narrow it down: block-major order, and nearby off-surface points through both paths
model.set_option('winding_tree', 1)
for b, vid in bad:
    model.set_option('winding_ray', 2); model.set_option('ray_pair_cap', 1)
    w1c = model.exterior_flags(verts, apply_segments=False, return_details=True)[1]
    model.set_option('ray_pair_cap', 16)
    print('  block-major order: ray %.6f' % w1c[b, vid].item())
    vb = verts[b]
    f = torch.tensor(faces[(faces == vid).any(1)], device=vb.device)
    n = torch.cross(vb[f[:, 1]] - vb[f[:, 0]], vb[f[:, 2]] - vb[f[:, 0]], dim=1).sum(0)
    n = n / n.norm()
    offs = torch.tensor([1e-2, 1e-3, 1e-4, 1e-5, -1e-5, -1e-4, -1e-3, -1e-2], device=vb.device)
    pts = (vb[vid][None] + offs[:, None] * n[None])[None].expand(verts.shape[0], -1, -1).contiguous()
    res = {}
    for mode in (0, 2):
        model.set_option('winding_ray', mode)
        w, _ = model.winding_points(verts, pts)
        res[mode] = w[b].cpu().numpy()
    print('  points along the normal, offsets', offs.tolist())
    print('    solid angle:', np.round(res[0], 4))
    print('    crossings  :', np.round(res[2], 4))
    # single-body call (other launch shape)
    model.set_option('winding_ray', 2)
    w_single = model.exterior_flags(verts[b:b + 1].contiguous(), apply_segments=False, return_details=True)[1]
    print('  single-body call: ray %.6f' % w_single[0, vid].item())
    ring = sorted(set(faces[(faces == vid).any(1)].ravel().tolist()) - {int(vid)})
    print('  ring', ring, 'coords of v', vb[vid].tolist())
    np.save(os.path.join(root, 'gpurun_out', 'diag_body.npy'), vb.cpu().numpy())
