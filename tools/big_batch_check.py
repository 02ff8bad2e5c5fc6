"""Ray-crossing flags against solid-angle flags at batch sizes that are not multiples of 8 / exceed one pass of the grids;
mismatching vertices must be ones that touch another triangle (on a jump of the winding number)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import torch, numpy as np
import test_gpu_contact as T
from helpers import touches_surface
for tag, batch in (('medium', 130), ('full', 70), ('small', 300)):
    g, verts = T._posed_batch(tag, batch, 5)
    model = T.make_model(g, T.golden_mask(tag), True, False)
    os.environ['TUCH_WINDING_RAY'] = '0'
    e0, w0 = model.exterior_flags(verts, apply_segments=False, return_details=True)[:2]
    es0 = model.exterior_flags(verts, apply_segments=True)
    os.environ['TUCH_WINDING_RAY'] = '2'
    e1, w1 = model.exterior_flags(verts, apply_segments=False, return_details=True)[:2]
    os.environ['TUCH_WINDING_RAY'] = '1'
    es1 = model.exterior_flags(verts, apply_segments=True)
    bad = torch.nonzero(e0 != e1).cpu().numpy()
    touching = sum(touches_surface(verts[b].cpu().numpy(), g['faces'], int(v)) for b, v in bad)
    print(tag, batch, 'body flag mismatches', len(bad), 'of which touching', touching, '| with segments', int((es0 != es1).sum()),
          '| max |dw| elsewhere %.2e' % float((w0 - w1).abs()[e0 == e1].max()))
os.environ.pop('TUCH_WINDING_RAY')
