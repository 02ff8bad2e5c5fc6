"""The step's middle (inside test + nearest-vertex search of the same vertices) as replayed graphs: each alone, one after the
other, side by side (what the step does).  Which of the two bounds the pair?   python tools/diag/middle_parts.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
from tuch_amd.smplify.losses import contact_model_for
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream(dev))
p = bench.build_problem(B, dev, 1002)
model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices.clone()


def timed(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cases = {
    'inside test alone (with the segment pass)': lambda: model.exterior_flags(verts),
    'inside test alone (body test only)': lambda: model.exterior_flags(verts, apply_segments=False),
    'search alone (iterative)': lambda: model.v2v_min(verts, iterative=True),
    'search alone (iterative, capped as beside the inside test)': lambda: model.v2v_min(verts, leave_room=True, iterative=True),
    'side by side': lambda: model.exterior_and_partner(verts, iterative=True),
}
res = {}
for name, fn in cases.items():
    res[name] = timed(bench.capture(fn, 3))
    print('%-62s %7.1f us' % (name, res[name]), flush=True)
model.set_option('overlap', 0)
res['one after the other'] = timed(bench.capture(lambda: model.exterior_and_partner(verts, iterative=True), 3))
print('%-62s %7.1f us' % ('one after the other', res['one after the other']))

# what each side costs beside the other: the pair with one side on fewer bodies
model.set_option('overlap', 1)
from tuch_amd.ops import _side_stream


def pair(bi, bs):
    vi, vs = verts[:bi].contiguous(), verts[:bs].contiguous()

    def fn():
        cur = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        side.wait_stream(cur)
        ext = model.exterior_flags(vi) if bi else None
        with torch.cuda.stream(side):
            out = model.v2v_min(vs, leave_room=True, iterative=True) if bs else None
        cur.wait_stream(side)
        return ext, out
    return timed(bench.capture(fn, 3))


print('side by side, bodies (inside test, search):')
for bi, bs in ((B, B), (B, B // 2), (B, 0), (B // 2, B), (0, B), (B // 2, B // 2)):
    print('   (%3d, %3d) %7.1f us' % (bi, bs, pair(bi, bs)), flush=True)
