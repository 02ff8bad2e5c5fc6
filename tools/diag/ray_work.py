"""Lane fill of the inside test's crossing kernel (batch 64): a tile is one leaf x up to 64 of the rays that pass its slabs;
(ray, element) pairs inside the slabs / (64 x element steps the wavefronts walk).   python tools/diag/ray_work.py [batch]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from tuch_amd import _C
from tuch_amd.ops import _workspace
from tuch_amd.smplify.losses import contact_model_for
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda:0')
p = bench.build_problem(B, dev, 1002)
model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices.clone()
L = _C.lib()
nbytes = L.tuch_exterior_workspace_bytes(model._handle, B)
ws = _workspace(nbytes, dev)
out = (ctypes.c_ulonglong * 4)()
_C.check(L.tuch_ray_work(model._handle, _C.ptr(verts), B, _C.ptr(ws), nbytes, out, _C.stream()))
steps, useful, listed, tiles = int(out[0]), int(out[1]), int(out[2]), int(out[3])
print('element steps walked by wavefronts %d (%.0f per body), tiles %d (%.0f per body, %.1f elements each)' % (steps, steps / B, tiles, tiles / B, steps / max(tiles, 1)))
print('(ray, element) pairs inside a listed leaf\'s slabs: %d = %.3f of the lanes issued leaf-major; block-major it would be %.3f' % (useful, useful / (64.0 * steps), useful / max(listed, 1)))
