"""Where the nearest-vertex scan's TIME goes across the chip (the search alone, batch 64): a second library with
-DTUCH_SCAN_CLOCKS stamps every wavefront's start and end (s_memrealtime, 100 MHz) and its place (XCC, SE, CU, SIMD).

    python tools/diag/scan_clocks.py build      # here: tuch_amd/libtuch_amd_scanclocks.so
    python tools/diag/scan_clocks.py [batch]    # on the GPU box
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tuch_amd', 'libtuch_amd_scanclocks.so')
if len(sys.argv) > 1 and sys.argv[1] == 'build':
    from tuch_amd import _build
    _build.build()
    objs = [os.path.join(_build.HERE, 'build', os.path.basename(s)[:-4] + '.o') for s in _build.sources()
            if not s.endswith('v2v.hip')]
    obj = os.path.join(_build.HERE, 'build', 'v2v_scanclocks.o')
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', _build.CSRC,
                    '-mllvm', '-amdgpu-mfma-vgpr-form', '-DTUCH_SCAN_CLOCKS', '-c', os.path.join(_build.CSRC, 'v2v.hip'), '-o', obj], check=True)
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + [obj], check=True)
    print(LIB)
    sys.exit(0)
os.environ['TUCH_AMD_LIB'] = LIB
import ctypes, numpy as np, torch, bench
from tuch_amd import _C
from tuch_amd.smplify.losses import contact_model_for
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iterative = (sys.argv[2] != '0') if len(sys.argv) > 2 else True
torch.cuda.set_stream(torch.cuda.Stream(dev))
p = bench.build_problem(B, dev, 1002)
model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
with torch.no_grad():
    verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices.clone()
for _ in range(6):
    model.v2v_min(verts, iterative=iterative)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    model.v2v_min(verts, iterative=iterative)
e1.record()
torch.cuda.synchronize()
print('search alone: %.1f us per call' % (e0.elapsed_time(e1) / 20 * 1e3))
L = _C.lib()
L.tuch_debug_scan_clocks.argtypes = [ctypes.c_void_p]
out = np.zeros((1 << 17, 8), np.uint64)
L.tuch_debug_scan_clocks(out.ctypes.data_as(ctypes.c_void_p))
c = out[out[:, 1] > 0]
# the last launch only: stamps within 1 ms of the latest end
end_all = c[:, 1].astype(np.int64).max()
c = c[(end_all - c[:, 1].astype(np.int64)) < 100000]
s = c[:, 0].astype(np.int64); e = c[:, 1].astype(np.int64)
t0 = s.min()
s = (s - t0) / 100.0; e = (e - t0) / 100.0
dur = e - s
hw = c[:, 2]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 15
h = (hw & np.uint64(0xffffffff)).astype(np.int64)
simd = (h >> 4) & 3; cu = (h >> 8) & 15; sh = (h >> 12) & 1; se = (h >> 13) & 7
by = (c[:, 3] >> np.uint64(32)).astype(np.int64)
cands = ((c[:, 3] >> np.uint64(8)) & np.uint64(0xffffff)).astype(np.int64)
trips = (c[:, 3] & np.uint64(0xff)).astype(np.int64)
ph = c[:, 4:8].astype(np.float64)
tot = ph[:, 0].sum()
print('a wavefront\'s life (s_memtime ticks): flushes of queued pairs %.1f %%, rows walked on the spot %.1f %%, leaf-per-lane tests '
      'with their loads %.1f %%, the rest (prologue, per-column tests of the candidates, queueing) %.1f %%'
      % (100 * ph[:, 1].sum() / tot, 100 * ph[:, 2].sum() / tot, 100 * ph[:, 3].sum() / tot,
         100 * (tot - ph[:, 1:].sum()) / tot))
print('%d wavefronts; kernel span %.1f us (first start -> last end); starts up to %.1f us' % (len(c), e.max(), s.max()))
print('wavefront lifetime: mean %.1f median %.1f p90 %.1f max %.1f us; sum %.0f us = %.2f wavefronts resident on average per SIMD (1024 SIMDs)'
      % (dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max(), dur.sum(), dur.sum() / e.max() / 1024))
print('trips per wavefront mean %.2f, candidates mean %.1f; lifetime vs candidates corr %.2f' % (trips.mean(), cands.mean(), np.corrcoef(dur, cands)[0, 1]))
# residency over time
for t in np.linspace(0, e.max(), 13)[:-1]:
    t1 = t + e.max() / 12
    res = (np.minimum(e, t1) - np.maximum(s, t)).clip(0).sum() / (t1 - t)
    print('  %6.1f..%6.1f us: %7.0f wavefronts resident (%.2f per SIMD), %5d started' % (t, t1, res, res / 1024, int(((s >= t) & (s < t1)).sum())))
# per-CU finishing times
key = xcc * 10000 + se * 1000 + sh * 100 + cu
ends = {}
work = {}
for k, ee, d in zip(key, e, dur):
    ends[k] = max(ends.get(k, 0), ee); work[k] = work.get(k, 0) + d
ev = np.array(list(ends.values())); wv = np.array(list(work.values()))
print('%d compute units seen; last end per CU: min %.1f median %.1f max %.1f us; wavefront-time per CU: min %.0f median %.0f max %.0f us'
      % (len(ev), ev.min(), np.median(ev), ev.max(), wv.min(), np.median(wv), wv.max()))
xe = [e[xcc == x].max() for x in sorted(set(xcc))]
print('last end per XCC:', ' '.join('%.1f' % v for v in xe))
# launch order against start time: is the heavy-first order kept?
o = np.argsort(s)
print('launch-order (blockIdx.y) of the first / last 5 %% started: %.0f / %.0f (of %d)' % (by[o[:len(o) // 20]].mean(), by[o[-len(o) // 20:]].mean(), by.max() + 1))
late = e > np.percentile(e, 99)
print('the 1 %% latest ends: lifetime mean %.1f us, start mean %.1f us, candidates mean %.1f' % (dur[late].mean(), s[late].mean(), cands[late].mean()))
# how good is the launch order?  list scheduling of the measured lifetimes over the resident slots, in launch order and
# longest-first; the means per tenth of the launch order
import heapq
# resident slots: the largest number of wavefronts alive at once
ev_t = np.concatenate([s, e]); ev_d = np.concatenate([np.ones_like(s), -np.ones_like(e)])
slots = int(np.cumsum(ev_d[np.argsort(ev_t, kind='stable')]).max())


def makespan(d, nslots):
    h = [0.0] * nslots
    heapq.heapify(h)
    for x in d:
        heapq.heappush(h, heapq.heappop(h) + x)
    return max(h)


order = np.lexsort((c[:, 3] & np.uint64(0), by))
bx = np.arange(len(c))
print('resident slots (most wavefronts alive at once): %d' % slots)
print('list scheduling of the measured lifetimes: launch order %.1f us, longest first %.1f us, perfect balance %.1f us'
      % (makespan(dur[np.argsort(by, kind="stable")], slots), makespan(np.sort(dur)[::-1], slots), dur.sum() / slots))
for q in range(10):
    sel = (by >= q * (by.max() + 1) / 10) & (by < (q + 1) * (by.max() + 1) / 10)
    print('  launch order tenth %d: lifetime mean %.1f us (max %.1f), candidates mean %.1f, start mean %.1f us' % (q, dur[sel].mean(), dur[sel].max(), cands[sel].mean(), s[sel].mean()))
# per (blockIdx.y) job over the bodies: how much of the variation is the job (static) and how much the body
jobs = {}
for y, d in zip(by, dur):
    jobs.setdefault(y, []).append(d)
jm = np.array([np.mean(v) for v in jobs.values()]); js = np.array([np.std(v) for v in jobs.values()])
print('per job over the bodies: mean of means %.1f us, spread of the means %.1f us, mean spread within a job %.1f us' % (jm.mean(), jm.std(), js.mean()))
# launch orders by job (blockIdx.y >> 1, every body the same): as launched, by mean lifetime, by mean candidates
job = by >> 1
nj = job.max() + 1
mean_life = np.array([dur[job == j].mean() for j in range(nj)])
mean_cand = np.array([cands[job == j].mean() for j in range(nj)])
bidx = np.arange(len(c))  # (the stamp slots are blockIdx.y * B + b: bodies of a job are launched together)
for name, keyv in (('as launched', -np.arange(nj)), ('by mean lifetime', mean_life), ('by mean candidates', mean_cand)):
    jorder = np.argsort(-keyv, kind='stable')
    rank = np.empty(nj, np.int64); rank[jorder] = np.arange(nj)
    o2 = np.argsort(rank[job] * 4 + (by & 1), kind='stable')
    print('list scheduling, jobs %-20s: %.1f us' % (name, makespan(dur[o2], slots)))
# static proxies of a job's weight from the mask alone (no geometry at model creation): leaves of the subtree the block's
# columns may pair with at all / only partly (the rim of the 30 cm geodesic neighbourhood: near in space too)
from tuch_amd.ops import cluster_tree
faces_np = model.faces_np.reshape(-1, 3)
V = int(faces_np.max()) + 1
F = faces_np.shape[0]
tr = cluster_tree(faces_np, V, max(32, F // 850))
nsub = nj // tr['frontier_off'][-1] if False else None
fo = tr['frontier_off']
fi = [k for k in range(len(fo) - 1) if (fo[k + 1] - fo[k]) * (len(tr['qperm']) // 128) == nj]
if fi:
    f0 = fo[fi[0]]
    qb_n = len(tr['qperm']) // 128
    order_tab = tr['launch_order'][f0 * qb_n: f0 * qb_n + nj]
    gm = p['geomask'].cpu().numpy().astype(bool)
    qp = tr['qperm'][:]
    nodes = tr['nodes']; rows = tr['rows']
    leaves = [i for i in range(len(nodes)) if nodes[i, 5] < 0]
    rim = np.zeros(nj); allowed = np.zeros(nj); pairs_allowed = np.zeros(nj); nleaves = np.zeros(nj)
    for j in range(nj):
        sub, qb = int(order_tab[j]) >> 16, int(order_tab[j]) & 0xffff
        node = tr['frontier_nodes'][f0 + sub]; skip = nodes[node, 4]
        cols = qp[qb * 128: qb * 128 + 128]
        cols = cols[:max(0, min(128, V - qb * 128))] if qb * 128 + 128 > V else cols
        for lf in leaves:
            if lf < node or lf >= skip: continue
            r0, rn = rows[lf]
            if rn == 0: continue
            sub_m = gm[np.ix_(cols, qp[r0:r0 + rn])]
            cnt = int(sub_m.sum())
            nleaves[j] += 1
            if cnt > 0:
                allowed[j] += 1; pairs_allowed[j] += cnt
                if cnt < sub_m.size: rim[j] += 1
    for name, x in (('leaves', nleaves), ('allowed leaves', allowed), ('rim leaves', rim), ('allowed pairs', pairs_allowed)):
        print('static proxy %-15s: corr with the mean lifetime of the job %.2f, with its mean candidates %.2f' % (name, np.corrcoef(x, mean_life)[0, 1], np.corrcoef(x, mean_cand)[0, 1]))
    for name, keyv in (('by rim leaves', rim), ('by rim, then allowed', rim * 1000 + allowed)):
        jorder = np.argsort(-keyv, kind='stable')
        rank = np.empty(nj, np.int64); rank[jorder] = np.arange(nj)
        o2 = np.argsort(rank[job] * 4 + (by & 1), kind='stable')
        print('list scheduling, jobs %-22s: %.1f us' % (name, makespan(dur[o2], slots)))
