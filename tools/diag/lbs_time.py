"""SMPL forward / backward alone: HIP-event time of the whole call at batch 64 and 8 (eager launches back to back and as one
captured graph), and a check against the CPU oracle's LBS on two bodies.

    python tools/diag/lbs_time.py            # prints forward / backward us; per-kernel times: run under rocprofv3 --kernel-trace --stats
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

dev = torch.device('cuda:0')
torch.cuda.set_stream(torch.cuda.Stream(device=dev))


def timed(fn, n=50):
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


for B in ([int(sys.argv[1])] if len(sys.argv) > 1 else (64, 8)):
    p = bench.build_problem(B, dev, 1002)
    bp = p['body_pose'].clone().requires_grad_(True)
    go = p['global_orient'].clone().requires_grad_(True)
    smpl = p['smpl']

    def fwd():
        with torch.no_grad():
            return smpl(global_orient=go, body_pose=bp, betas=p['betas'])
    out = smpl(global_orient=go, body_pose=bp, betas=p['betas'])
    gv = torch.randn_like(out.vertices)
    gj = torch.randn_like(out.joints)

    def fwd_bwd():
        o = smpl(global_orient=go, body_pose=bp, betas=p['betas'])
        bp.grad = go.grad = None
        torch.autograd.backward([o.vertices, o.joints], [gv, gj])
    g_fwd = bench.capture(fwd, 3)
    g_all = bench.capture(fwd_bwd, 3)
    tf, ta = timed(g_fwd), timed(g_all)
    print('batch %2d: forward %6.1f us   forward + backward %6.1f us   (backward %6.1f)   [graph replays]' % (B, tf, ta, ta - tf))
    if B == 64:
        from oracle import lbs as ol
        m = ol.model_tensors(p['body'], torch.float64)
        o = fwd()
        for b in (0, 37):
            v, j = ol.smpl_forward(m, p['betas'][b:b + 1].cpu().double(), bp[b:b + 1].detach().cpu().double(),
                                   go[b:b + 1].detach().cpu().double())
            ev = float((o.vertices[b].cpu().double() - v[0]).abs().max())
            print('   body %d: max |verts - oracle(fp64)| = %.2e m' % (b, ev))
