#!/usr/bin/env python3
"""Headline benchmark: SMPLify-DC stage-2 fit iterations per second at batch 64 per GPU.

One *step* = one pass of the reference's stage-2 loop body (tuch/smplify/smplifydc.py:155-183)
over one batch: SMPL forward -> contact_fitting_loss (reprojection + GMM prior + self-contact
push/pull terms with winding-number inside test and segment filter + region-to-region term)
-> backward -> Adam step, on SMPL-sized synthetic bodies (V=6890, F=13776).

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1: spawns N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The batch dimension shards across ranks.  --gpus N > 1 defaults to STRONG scaling at the global batch of 64 that
SURVEY 8(e) / north_star name (64/N bodies per GPU; the weak-scaling figure, 64 bodies per GPU, is timed in the same run and
reported as `weak_scaling`; --weak makes it the headline, --global-batch G picks another total).  The fits of different bodies never exchange data, so there is no collective inside
the loop: the two floats [sum of losses, body count] are all-reduced (RCCL) once per timed block of K steps.
Rank 0 prints ONE JSON line (contract in the task description) including
  roofline     -- dominant kernel of the step (the masked vertex-distance search, v2v_scan_kernel): `frac` = EXECUTED vector
                  work (SQ_INSTS_VALU of the committed PMC pass x 64 lanes x 2 / launch time measured here with HIP events /
                  FP32 vector peak); `equivalent_frac` = SURVEY 8(d)'s all-pairs figure (most pairs are pruned); valu_busy /
                  traffic parsed from the PMC summaries committed under profiles/ (the newest round's files)
  roofline_inside_test -- the second kernel group (inside test by ray crossings): executed operations per launch
  step_equivalent_x_peak -- the whole step priced in SURVEY 8(d)'s pair units over the FP32 vector peak (> 1: pruning)
  weak_scaling -- N > 1 only: the same step with 64 bodies on every rank (the headline of N > 1 is strong scaling at
                  global batch 64)
  cpu_baseline -- the CPU oracle (test infrastructure) timed on this box's host cores, rank 0, N=1
  selfcheck    -- after the timed blocks: the objective the replayed graph reports against an eager evaluation at the
                  same parameters, and two sampled bodies of that state against the CPU oracle
  rccl_smoke   -- N=1 only, in a child process: init_process_group('nccl', world_size=1), a device all-reduce, a
                  contact loss with the all-reduced valid count captured in a hipGraph and replayed, destroy
  kernels_per_step -- launches in one captured step (the graph's kernel nodes)
  repeat_ms_per_step -- the same K-step block timed --repeats times (median / min / max)
  shard_sweep  -- the step at 8/16/32/64 bodies on one GPU (the per-GPU shards of a global batch of 64)
  workloads    -- the per-rank workloads of BASELINE configs 3, 4 and 5
  worst_case   -- every body self-penetrating; folded: limbs pushed THROUGH the body
  float_atomics_mode -- the step with TUCH_DETERMINISTIC=0 (the headline runs deterministic: bit-reproducible gradient scatters)
  irregular_topology -- the step and its pruning statistics on the irregular-topology body next to the lat-long one
--config {2,3,4-shard,5-shard} makes one of those workloads the timed step instead (its own metric name).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

BATCH_PER_GPU = 64
FLOP_PER_WINDING_PAIR = 67        # 63 arithmetic + 3 sqrt + 1 atan2
FLOP_PER_V2V_PAIR = 8
# executed arithmetic of the strip walk per (query, stream element): 3 sub, |.|^2 (5) + sqrt, two dot products
# (10), numerator (5), denominator (8), small-angle atan (rcp + 8)
FLOP_PER_STRIP_ELEMENT = 41
# executed arithmetic of the ray-crossing walk per (ray, strip element): 3 sub, 2 edge functions (4 mul + 2 sub, never
# contracted), depth determinant (1 mul + 2 fma = 5), min3 + max3 (2 + 2), min, max, 1 mul for the tie test = 21 FP32
# operations in 20 VALU instructions since round 4 (packed subtraction and packed edge-function products, the count as one
# multiply-add; 21.0 = SQ_INSTS_VALU / element with the kernel's prologues, profiles/r04_z_pmc_sq.txt; round 2: 29)
OPS_PER_RAY_ELEMENT = 21
VALU_INSTR_PER_RAY_ELEMENT = 21
PEAK_FP32_VECTOR_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs x 16 lanes x 2 (FMA) x 2 (packed) x 2.4 GHz
# what a stream of plain (non-packed) wave64 instructions can issue: 256 x 4 SIMDs x 16 lanes x 2.4 GHz lane-instructions/s
# (one wave64 VALU instruction occupies its SIMD for 4 cycles: 4.1-4.5 measured, tools/ubench/valu_rate.hip -fno-slp-vectorize)
PEAK_PLAIN_ISSUE_TLANEOPS = 39.3
PEAK_HBM_GBS = 8000.0


def _pmc_rows(path):
    """{first 60 characters of the kernel name: {counter: average per launch}} of a scripts/rocprof_pmc_summary.py file."""
    rows = {}
    with open(path) as f:
        for line in f:
            if line.startswith('#') or '=' not in line:
                continue
            counters = {}
            for field in line.split():
                name, eq, value = field.partition('=')
                if eq and name.isupper():
                    try:
                        counters[name] = float(value)
                    except ValueError:
                        pass
            rows[line[:60].strip()] = counters
    return rows


def _source_hash(name):
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, 'tuch_amd', 'csrc', name), 'rb').read()).hexdigest()[:16]
    except OSError:
        return None


def profile_staleness(sq_file, sources):
    """(stale, reason): were the committed counters profiled on the kernel sources this run executes?  The PMC summaries
    carry '# source sha256: file=hash ...' of the tree they were collected from (scripts/rocprof_pmc_summary.py); a
    summary without that line (rounds 1-4) cannot be verified and counts as stale."""
    recorded = {}
    with open(sq_file) as f:
        for line in f:
            if line.startswith('# source sha256:'):
                recorded = dict(item.split('=', 1) for item in line.split(':', 1)[1].split() if '=' in item)
                break
    if not recorded:
        return True, 'no source hashes recorded in %s' % os.path.basename(sq_file)
    changed = [s for s in sources if recorded.get(s) != _source_hash(s)]
    if changed:
        return True, '%s changed since %s was collected' % (', '.join(changed), os.path.basename(sq_file))
    return False, None


def live_lane_fraction():
    """(fraction, source) of the search's row arithmetic: lanes with a (leaf, column) pair in reach / lanes issued, from the
    newest committed counting run (tools/diag/scan_counts.py: a second library built with -DTUCH_SCAN_COUNTS); from
    profiles/, NOT measured in this run."""
    import glob
    import re
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_scan_counts.txt')), reverse=True):
        m = re.search(r'^live_lane_fraction ([\d.]+)', open(f).read(), re.M)
        if m:
            return float(m.group(1)), 'profiles/' + os.path.basename(f)
    return None, None


def profile_constants():
    """HBM-side bytes per launch and VALU-busy fraction of the two big kernels at batch 64, parsed at start-up from the
    newest committed PMC summaries (profiles/rNN_x_pmc_{fetch,write,sq}.txt: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE /
    SQ_* in separate passes, KB x 1024; VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)).
    From profiles/, NOT measured in this run."""
    import glob
    import re
    out = {}
    tags = sorted({re.match(r'(r\d+_[a-z]+)_pmc_sq\.txt', os.path.basename(f)).group(1)
                   for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_sq.txt'))
                   if re.match(r'(r\d+_[a-z]+)_pmc_sq\.txt', os.path.basename(f))})
    for tag in reversed(tags):
        files = {k: os.path.join(ROOT, 'profiles', '%s_pmc_%s.txt' % (tag, k)) for k in ('fetch', 'write', 'sq')}
        if not all(os.path.exists(f) for f in files.values()):
            continue
        rows = {k: _pmc_rows(f) for k, f in files.items()}
        for key, names in (('search', ('v2v_scan_shared_kernel', 'v2v_scan_kernel', 'v2v_tree_kernel')), ('ray_leaf_kernel', ('ray_leaf_kernel',))):
            pick = lambda table: next(((k, c) for n in names for k, c in table.items() if n in k), (None, None))
            (kname, fe), (_, wr), (_, sq) = pick(rows['fetch']), pick(rows['write']), pick(rows['sq'])
            if fe and wr and sq and sq.get('GRBM_GUI_ACTIVE'):
                stale, why = profile_staleness(files['sq'], ('v2v.hip', 'common.h', 'tree_device.h') if key == 'search' else
                                               ('ray_winding.hip', 'common.h', 'tree_device.h'))
                out[key] = {'kernel': (re.search(r'(\w+_kernel)', kname) or [None, kname])[1], 'stale': stale, 'stale_reason': why,
                            'traffic_bytes': int((fe['FETCH_SIZE'] + wr['WRITE_SIZE']) * 1024),
                            'valu_busy': round(sq['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * sq['GRBM_GUI_ACTIVE'] / 8), 3),
                            'valu_instr': sq.get('SQ_INSTS_VALU'), 'salu_instr': sq.get('SQ_INSTS_SALU'),
                            'source': 'profiles/%s_pmc_{fetch,write,sq}.txt (batch 64, inside the step); from profiles/, '
                                      'not measured in this run' % tag}
        if len(out) == 2:
            break
    for key in ('search', 'ray_leaf_kernel'):
        out.setdefault(key, {'kernel': None, 'traffic_bytes': None, 'valu_busy': None, 'valu_instr': None, 'salu_instr': None,
                             'stale': True, 'stale_reason': 'no PMC summary under profiles/',
                             'source': 'no PMC summary under profiles/'})
    return out


def timeline_constants():
    """Head / middle / tail of ONE replayed step from the newest committed timelines (profiles/rNN_x_graph_timeline.txt,
    _b8.txt: tools/graph_timeline.py under rocprofv3 --kernel-trace): head = up to the start of the nearest-vertex
    search's big kernel, middle = until both big kernels (search, ray_leaf_kernel) are done, tail = the rest.  From
    profiles/, NOT measured in this run (the profiler adds ~4 % to the step)."""
    import glob
    import re
    out = {}
    for key, suffix in (('batch64', '_graph_timeline.txt'), ('batch8', '_graph_timeline_b8.txt')):
        files = sorted(f for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*' + suffix))
                       if re.match(r'r\d+_[a-z]+' + re.escape(suffix) + '$', os.path.basename(f)))
        if not files:
            continue
        rows, head = [], {}
        with open(files[-1]) as f:
            for line in f:
                m = re.match(r'kernels in the step: (\d+), wall ([\d.]+) us, summed kernel time ([\d.]+) us', line)
                if m:
                    head = {'kernels': int(m.group(1)), 'wall_us': float(m.group(2)), 'summed_kernel_us': float(m.group(3))}
                m = re.match(r'\s*([\d.]+)\s+([\d.]+)\s+(.*)', line)
                if m and not line.startswith('kernels') and not line.startswith('time'):
                    rows.append((float(m.group(1)), float(m.group(2)), m.group(3)))
        big = [r for r in rows if re.search(r'v2v_(scan|mfma|leaves|tree)\w*_kernel(<[^>]*>)?\(', r[2]) and 'finalize' not in r[2]
               or 'ray_leaf_kernel' in r[2]]
        search = [r for r in big if 'v2v_' in r[2]]
        if head and big and search:
            mid_end = max(r[0] + r[1] for r in big)
            out[key] = dict(head, head_us=round(search[0][0], 1), middle_us=round(mid_end - search[0][0], 1),
                            tail_us=round(head['wall_us'] - mid_end, 1), source='profiles/' + os.path.basename(files[-1]))
    return out


_BODY = {}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='bodies per GPU (weak scaling)')
    ap.add_argument('--global-batch', type=int, default=None,
                    help='strong scaling: total bodies, split evenly over the GPUs (the default for --gpus N > 1: 64, '
                         'SURVEY 8(e); the weak-scaling figure at 64 bodies per GPU is timed next to it)')
    ap.add_argument('--weak', action='store_true',
                    help='N > 1: weak scaling (--batch bodies per GPU) as the headline instead of strong scaling at '
                         'global batch 64')
    ap.add_argument('--repeats', type=int, default=5, help='how many times the K-step block is timed')
    ap.add_argument('--config', default='2', choices=['2', '3', '4-shard', '5-shard'],
                    help='which BASELINE config is the timed step (default 2 = the headline)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--eager', action='store_true', help='launch every step eagerly instead of replaying a hipGraph')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-torch-chain', action='store_true',
                    help='skip the torch-CPU op-chain baseline (one body, ~8 GB of host memory, ~30 s)')
    ap.add_argument('--no-extras', action='store_true', help='skip shard sweep, workloads, worst case, contact-loss eval')
    ap.add_argument('--rccl-smoke', action='store_true', help='(internal) run the one-rank RCCL smoke and exit')
    ap.add_argument('--no-rccl-smoke', action='store_true')
    ap.add_argument('--allreduce-per-block', action='store_true',
                    help='N > 1: reduce the two floats of statistics once per timed block instead of once per step (the default '
                         'for N > 1 is one all-reduce per step, SURVEY 8e; the other form is timed as well and reported beside it)')
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    """--gpus N without a launcher: re-exec under torch.distributed.run, one rank per GPU of this node."""
    backend = os.environ.get('TUCH_BENCH_BACKEND', 'nccl')
    if backend == 'nccl' and torch.cuda.device_count() < args.gpus:
        raise SystemExit('bench.py --gpus %d: only %d GPU(s) visible (RCCL needs one device per rank)'
                         % (args.gpus, torch.cuda.device_count()))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def synthetic_body(topology='uv'):
    """'uv': the lat-long sphere warped into a humanoid (V=6890, F=13776: the headline); 'ico': the irregular one
    (geodesic icosahedron + edge flips: V=6762, valence 4-9, painted ragged segments)."""
    from synthetic import make_body
    key = 'body' if topology == 'uv' else 'body_' + topology
    if key not in _BODY:
        _BODY[key] = make_body(84, 82, seed=1234) if topology == 'uv' else make_body(topology='ico', freq=26, seed=1234)
    return _BODY[key]


def build_problem(batch, device, seed, penetrating_fraction=0.5, folded=False, topology='uv'):
    from tuch_amd.models.smpl import SMPL
    from tuch_amd.smplify.prior import MaxMixturePrior
    from synthetic import folded_poses, random_poses
    from tuch_amd.utils.geometry import perspective_projection
    from tuch_amd.utils.segmentation import BatchBodySegment
    body = synthetic_body(topology)
    key = ('shared', str(device), topology)
    if key not in _BODY:       # model constants are shared by every problem built on this device
        smpl = SMPL(model_data=body, batch_size=batch).to(device)
        prior = MaxMixturePrior(num_gaussians=8, gmm=body.gmm).to(device)
        face_row = torch.tensor(body.faces, dtype=torch.long, device=device)
        geomask = torch.tensor(body.geodesics > 0.3, device=device)
        segments = BatchBodySegment(list(body.segments.keys()), face_row, body.segments)
        cdict = {'classes': [list(p) for p in body.region_pairs], 'csig': dict(body.regions)}
        _BODY[key] = (smpl, prior, face_row, geomask, segments, cdict)
    smpl, prior, face_row, geomask, segments, cdict = _BODY[key]
    face_tensor = face_row[None].expand(batch, -1, -1)
    rng = np.random.Generator(np.random.PCG64(seed))
    bp, go, be = folded_poses(batch, seed) if folded else random_poses(batch, seed, penetrating_fraction)
    t = lambda a: torch.tensor(a, device=device)
    body_pose, global_orient, betas = t(bp), t(go), t(be)
    cam_t = torch.tensor([[0.0, 0.0, 20.0]], device=device).repeat(batch, 1)
    cam_c = torch.zeros(batch, 2, device=device)
    with torch.no_grad():
        tgt = smpl(global_orient=global_orient, body_pose=body_pose + 0.05 * torch.randn_like(body_pose),
                   betas=betas)
        j2d = perspective_projection(tgt.joints, torch.eye(3, device=device)[None].expand(batch, -1, -1),
                                     cam_t, 5000., cam_c)
    j2d = j2d + 2.0 * t(rng.standard_normal((batch, 49, 2)).astype(np.float32))
    conf = t((0.5 + 0.5 * rng.random((batch, 49))).astype(np.float32))
    conf[:, [1, 9, 12, 27, 28]] = 0.0                                   # smplifydc.py:153
    gt = t((rng.random((batch, len(body.region_pairs))) < 0.03).astype(np.float32))
    return dict(body=body, smpl=smpl, prior=prior, face_tensor=face_tensor, geomask=geomask,
                segments=segments, cdict=cdict, body_pose=body_pose, global_orient=global_orient,
                betas=betas, cam_t=cam_t, cam_c=cam_c, j2d=j2d, conf=conf, gt=gt,
                ignore=torch.zeros(batch, dtype=torch.bool, device=device),
                has_dc=torch.ones(batch, dtype=torch.bool, device=device))


def make_step(p):
    """The stage-2 loop body of SMPLifyDC.__call__ (smplifydc.py:155-183); returns [loss sum, bodies]."""
    from tuch_amd import ops
    from tuch_amd.smplify.losses import contact_fitting_loss
    body_pose = p['body_pose'].clone().requires_grad_(True)
    global_orient = p['global_orient'].clone().requires_grad_(True)
    from tuch_amd.optim import make_adam
    # the reference's torch.optim.Adam update as one launch (tuch_amd/optim.py, what SMPLifyDC's own loops use)
    # (fuse_backward: the body model's last backward kernel applies the update itself, as in SMPLifyDC's kept stage-2 loop)
    opt = make_adam([body_pose, global_orient], 1e-2, fuse_backward=os.environ.get('TUCH_FUSED_ADAM', '1') != '0')
    count = torch.full((), float(body_pose.shape[0]), device=body_pose.device)

    def step():
        out = p['smpl'](global_orient=global_orient, body_pose=body_pose, betas=p['betas'])
        loss = contact_fitting_loss(body_pose, global_orient, None, None, p['betas'], out.joints,
                                    p['geomask'], 0.02, p['cam_t'], p['cam_c'], p['j2d'], p['conf'],
                                    p['prior'], cdict=p['cdict'], gt_contact=[p['gt'], None],
                                    ignore_idxs=p['ignore'], has_discrete_contact=p['has_dc'],
                                    verts=out.vertices, face_tensor=p['face_tensor'],
                                    focal_length=5000., contact_loss_weight=2000.0,
                                    segments=p['segments'])
        opt.zero_grad(set_to_none=True)
        ops.backward_scalar(loss)                 # as SMPLifyDC._Stage._one does
        opt.step()
        return loss.detach(), count               # [loss sum, bodies]: stacked by the caller, once per timed block

    def objective():
        """The stage-2 objective at the CURRENT parameters, eager launches, no update; also the posed vertices."""
        with torch.no_grad():
            out = p['smpl'](global_orient=global_orient, body_pose=body_pose, betas=p['betas'])
            loss = contact_fitting_loss(body_pose, global_orient, None, None, p['betas'], out.joints,
                                        p['geomask'], 0.02, p['cam_t'], p['cam_c'], p['j2d'], p['conf'],
                                        p['prior'], cdict=p['cdict'], gt_contact=[p['gt'], None],
                                        ignore_idxs=p['ignore'], has_discrete_contact=p['has_dc'],
                                        verts=out.vertices, face_tensor=p['face_tensor'],
                                        focal_length=5000., contact_loss_weight=2000.0,
                                        segments=p['segments'])
        return float(loss), out.vertices.detach(), out.joints.detach(), body_pose.detach().clone()
    step.objective = objective
    return step


def make_fit(p, iters):
    """BASELINE configs[2]: one demo_smplify_dc.py-style fit = `iters` stage-1 + `iters` stage-2 iterations."""
    from tuch_amd.smplify.smplifydc import SMPLifyDC
    batch = p['body_pose'].shape[0]
    dev = p['body_pose'].device
    fitter = SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=iters, focal_length=5000.,
                       geodistssmpl=torch.tensor(p['body'].geodesics, device=dev), geothres=0.3, euclthres=0.02,
                       device=dev, smpl=p['smpl'], pose_prior=p['prior'])
    kp = torch.cat([p['j2d'], p['conf'][..., None]], 2)
    init_pose = torch.cat([p['global_orient'], p['body_pose']], 1)
    stats = torch.zeros(2, device=dev)
    stats[1] = float(batch)

    def fit():
        res = fitter(init_pose, p['betas'], p['cam_t'], p['cam_c'], kp, use_contact=True, contactlist=p['cdict'],
                     gt_contact=[p['gt'], None], ignore_idxs=p['ignore'], has_discrete_contact=p['has_dc'],
                     contact_loss_weight=2000.0, segments=p['segments'])
        stats[0] = res[5].sum()
        return stats
    return fit, fitter


def regressor_loss(p, use_hd):
    import types
    from tuch_amd.train.loss import RegressorLoss
    body = p['body']
    dev = p['body_pose'].device
    key = ('crit', str(dev), use_hd)
    if key not in _BODY:
        opts = types.SimpleNamespace(contact_loss_weight=1.0, shape_loss_weight=0.5, keypoint_loss_weight=5.0, pose_loss_weight=1.0,
                                     beta_loss_weight=0.001, openpose_train_weight=0.0, gt_train_weight=1.0)
        _BODY[key] = RegressorLoss(opts, dev, body.num_verts, p['face_tensor'],
                                   torch.tensor(body.geodesics, device=dev), geothres=0.3, euclthres=0.02,
                                   face_tensor=p['face_tensor'], use_hd=use_hd, segments=p['segments'],
                                   hd_regressor=(body.hd_bary_idx, body.hd_bary_w), hd_faces=body.hd_face_id)
    return _BODY[key]


FRESH_BATCHES = 4      # distinct pose batches rotated through the "fresh bodies" legs (a training loop never repeats a body)


def fresh_problems(p, device, seed):
    """p followed by FRESH_BATCHES - 1 more problems of the same batch size with other random poses / shapes."""
    key = ('fresh', str(device), p['body_pose'].shape[0], seed)
    if key not in _BODY:
        _BODY[key] = [build_problem(p['body_pose'].shape[0], device, seed + 101 * k) for k in range(1, FRESH_BATCHES)]
    return [p] + _BODY[key]


def make_train_step(p, use_hd, smplify_iters=0, fresh=None):
    """The contact part of a train.py-style step in isolation (used for the HD / plain comparison): SMPL forward with
    pose2rot=False (train_module.py:202-204) -> RegressorLoss.contact_loss (mean over the valid bodies of ALL ranks)
    -> backward to the rotation matrices and betas.
    fresh: a list of problems; step.next_bodies() then writes the NEXT problem's rotation matrices and betas into the
    step's input tensors in place (a copy on the stream, outside the captured step) -- what a training loop does to a
    captured step between replays: every replay sees bodies it has never seen."""
    from tuch_amd.utils.geometry import batch_rodrigues
    batch = p['body_pose'].shape[0]
    dev = p['body_pose'].device
    crit = regressor_loss(p, use_hd)

    def inputs(q):
        full_pose = torch.cat([q['global_orient'], q['body_pose']], dim=1).detach()
        return batch_rodrigues(full_pose.reshape(-1, 3)).view(batch, 24, 3, 3).detach().clone(), q['betas'].detach().clone()
    rot0, bet0 = inputs(p)
    rotmat = rot0.clone().requires_grad_(True)
    betas = bet0.clone().requires_grad_(True)
    valid = torch.ones(batch, dtype=torch.bool, device=dev)
    stats = torch.zeros(2, device=dev)
    stats[1] = float(batch)

    def step():
        rotmat.grad = betas.grad = None
        o = p['smpl'](betas=betas, body_pose=rotmat[:, 1:], global_orient=rotmat[:, :1], pose2rot=False)
        loss = crit.contact_loss(o.vertices, valid)
        loss.backward()
        stats[0] = loss.detach() * batch
        return stats
    if fresh:
        pool = [inputs(q) for q in fresh]
        state = {'i': 0}

        def next_bodies():
            state['i'] = (state['i'] + 1) % len(pool)
            with torch.no_grad():
                rotmat.copy_(pool[state['i']][0])
                betas.copy_(pool[state['i']][1])
        step.next_bodies = next_bodies
    return step


def make_tuch_step(p, run_smplify, smplify_iters=10, seed=77):
    """BASELINE configs[3] / [4] per-rank step: the whole TUCH.forward_train_step (tuch/train/train_module.py:105-335,
    restated in tuch_amd/train/train_module.py and pinned to the reference's own output by tests/test_gpu_train_step.py)
    + backward.  The HMR / SPIN regressors are small deterministic stand-ins (synthetic.make_regressor): the
    ResNet-50 is stock PyTorch and not part of the path; everything downstream of its output is the real step --
    SMPL with rotation matrices, rotation matrix -> axis-angle, estimate_translation, the dictionary of best fits,
    contact_from_verts, [SMPLify-DC in the loop with contact], RegressorLoss with the HD contact term."""
    import tempfile
    import types
    from tuch_amd.smplify.smplifydc import SMPLifyDC
    from synthetic import make_regressor, make_train_batch
    from tuch_amd.train.train_module import TUCH
    batch = p['body_pose'].shape[0]
    dev = p['body_pose'].device
    body = p['body']
    options = types.SimpleNamespace(
        batch_size=batch, img_res=224, run_smplify=run_smplify, use_contact_in_the_loop=True,
        contact_in_the_loop_loss_weight=2000.0, smplify_threshold=100.0, num_smplify_iters=smplify_iters,
        contact_loss_weight=1.0, shape_loss_weight=0.5, keypoint_loss_weight=5.0, pose_loss_weight=1.0, beta_loss_weight=0.001,
        openpose_train_weight=0.0, gt_train_weight=1.0, checkpoint_dir=tempfile.mkdtemp(prefix='tuch_bench_'))
    datasets = (('dsA', 4096), ('dsB', 4096))
    train_ds = types.SimpleNamespace(dataset_dict={n: i for i, (n, _) in enumerate(datasets)},
                                     datasets=[range(k) for _, k in datasets])
    smplify = SMPLifyDC(step_size=1e-2, batch_size=batch, num_iters=smplify_iters, focal_length=5000.,
                        geodistssmpl=torch.tensor(body.geodesics, device=dev), geothres=0.3, euclthres=0.02, device=dev,
                        smpl=p['smpl'], pose_prior=p['prior'])
    module = TUCH(options=options, device=dev, datasets=(train_ds, None), bodymodel=p['smpl'],
                  spin_model=make_regressor(11).to(dev), regressor=make_regressor(12).to(dev), optimization=smplify,
                  criterion=regressor_loss(p, True), geodistssmpl=None, contactlists=p['cdict'])
    raw = make_train_batch(body, batch, seed, datasets)
    input_batch = {k: (torch.tensor(v, device=dev) if not isinstance(v, list) else v) for k, v in raw.items()}
    params = [q for q in module.model.parameters()]
    stats = torch.zeros(2, device=dev)
    stats[1] = float(batch)

    def step():
        for q in params:
            q.grad = None
        loss, _, _ = module.forward_train_step(input_batch)
        loss.backward()
        stats[0] = loss.detach() * batch
        return stats
    # fresh bodies: the images (what the regressor stand-in turns into poses and shapes), keypoints and pseudo ground truth of
    # FRESH_BATCHES different input batches, written into the step's input tensors in place between calls / replays (dataset
    # names, sample indices and the has_* flags stay: they steer the fits dictionary, not the contact path)
    swap = ('img', 'keypoints', 'pose_3d', 'pose', 'betas', 'contact_vec')
    pool = [{k: input_batch[k].clone() for k in swap}]
    for k in range(1, FRESH_BATCHES):
        other = make_train_batch(body, batch, seed + 101 * k, datasets)
        pool.append({name: torch.tensor(other[name], device=dev) for name in swap})
    state = {'i': 0}

    def next_bodies():
        state['i'] = (state['i'] + 1) % len(pool)
        with torch.no_grad():
            for name in swap:
                input_batch[name].copy_(pool[state['i']][name])
    step.next_bodies = next_bodies
    step.module, step.inputs = module, input_batch
    return step


def check_tuch_step(p, step, run_smplify, sample=(0, 17)):
    """The per-rank config-4 / config-5 step at FULL size against the CPU oracle (the checker -- tests/ infrastructure, not
    the thing measured): one eager ``forward_train_step`` with a spy on ``RegressorLoss.contact_loss`` (and on the
    SMPLify-DC call in the loop); for two sampled bodies the HD contact loss of the regressor's vertices against
    ``oracle.contact.train_contact_body`` and, with SMPLify-DC in the loop, the stage-2 contact value of the FITTED
    vertices against ``oracle.contact.smplify_contact_body``."""
    from oracle import contact as oc
    from tuch_amd.ops import MODE_SMPLIFY, contact_terms
    from tuch_amd.smplify.losses import contact_model_for
    module = step.module
    crit = module.criterion_cospin
    body = p['body']
    seen = {}
    real = crit.contact_loss

    def spy(verts, valid):
        seen['verts'], seen['valid'] = verts.detach(), valid.detach().clone()
        return real(verts, valid)

    class Spy:                                   # the SMPLify-DC object in the loop: its call's vertices are kept
        def __init__(self, inner):
            self.inner = inner

        def __call__(self, *a, **k):
            out = self.inner(*a, **k)
            seen['fit_verts'] = out[0].detach()
            return out

        def __getattr__(self, name):
            return getattr(self.inner, name)
    crit.contact_loss = spy
    inner = module.smplify
    module.smplify = Spy(inner)
    try:
        loss, losses, _ = module.forward_train_step(step.inputs)
        torch.cuda.synchronize()
    finally:
        crit.contact_loss = real
        module.smplify = inner
    gm = body.geodesics > 0.3
    segs = [oc.Segment(n, body.faces, sg['vidx'], list(sg['bands'].values())) for n, sg in body.segments.items()]
    one = torch.ones(1, dtype=torch.bool, device=seen['verts'].device)
    sample = [b for b in sample if b < seen['verts'].shape[0]]
    worst_train = worst_fit = 0.0
    for b in sample:
        with torch.no_grad():
            got = float(real(seen['verts'][b:b + 1], one))
        want = float(oc.train_contact_body(seen['verts'][b].cpu().numpy(), body.faces, gm, 0.02, segs, True, hd_idx=body.hd_bary_idx,
                                           hd_w=body.hd_bary_w, hd_face=body.hd_face_id)['loss'])
        worst_train = max(worst_train, abs(got - want) / max(abs(want), 1e-9))
        if run_smplify and 'fit_verts' in seen:
            model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
            v = seen['fit_verts'][b:b + 1].contiguous()
            ext, _, partner, _ = model.exterior_and_partner(v, apply_segments=True)
            per_body, _ = contact_terms(v, partner, ext, None, MODE_SMPLIFY, 0.02)
            r = oc.smplify_contact_body(v[0].cpu().numpy(), body.faces, gm, 0.02, segs, None)
            worst_fit = max(worst_fit, abs(float(per_body[0]) - r['contact']) / max(abs(r['contact']), 1e-9))
    out = {'oracle_bodies': sample, 'loss': float(loss), 'loss_contact': float(losses['loss_contact']),
           'hd_contact_loss_max_rel_err_vs_oracle': worst_train, 'valid_bodies': int(seen['valid'].sum())}
    if run_smplify:
        out['fitted_vertices_contact_value_max_rel_err_vs_oracle'] = worst_fit
    out['ok'] = bool(np.isfinite(out['loss']) and worst_train < 1e-4 and worst_fit < 1e-4)
    return out


def capture(step, warmup):
    """Capture one whole step (forward, backward, Adam) into a hipGraph; returns the replay callable."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(max(warmup, 3)):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode='thread_local'):
        out = step()

    def replay():
        graph.replay()
        return out
    replay.objective = getattr(step, 'objective', None)
    replay.next_bodies = getattr(step, 'next_bodies', None)
    replay.graph = graph
    # the graph holds raw addresses of everything `step` owns (parameters, Adam state, ...): they must live as
    # long as the graph does.  (Round 1 dropped `step` here; its tensors were then recycled by the next regular
    # allocation -- the all-reduce's clone for N > 1 -- which is what "faulted next to a process group".)
    replay.keep_alive = (step, graph)
    return replay


def time_kernel(fn, iters, between=None):
    """Seconds per call (HIP events on the current stream); best of two passes after two warm-up calls, so that
    a one-off allocator growth does not land in the figure.
    between: called before every timed call (the "fresh bodies" legs: new inputs written in place; its few small device
    copies are part of the figure)."""
    call = fn if between is None else (lambda: (between(), fn()))
    for _ in range(2 if between is None else FRESH_BATCHES + 1):
        call()
    torch.cuda.synchronize()

    def best_of_two(f):
        best = float('inf')
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters * 1e-3)
        return best
    return best_of_two(call)


def rooflines(p, batch):
    """The two kernel groups that dominate the step, measured live (HIP events on the launch stream).

    roofline: the masked vertex-to-vertex search (v2v_tree_kernel + its row / box / seed / finalize helpers), since the
    inside test went to ray crossings the largest kernel of the step.  `achieved` is SURVEY.md 8(d)'s algorithmic work
    (8 FLOP per ordered vertex pair, V^2 pairs per body) over the launch time; the kernel prunes ~60 % of the rows by
    box distance, so the executed arithmetic is lower -- VALU-busy from the committed PMC pass says how full the
    vector units are.
    roofline_inside_test: exterior flags by signed ray crossings (csrc/ray_winding.hip), priced on EXECUTED operations:
    21 plain FP32 operations per (ray, strip element), none of them packed, two of them FMAs."""
    from tuch_amd.smplify.losses import contact_model_for
    body = p['body']
    v, f = body.num_verts, body.num_faces
    model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
    with torch.no_grad():
        verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices
    # ---- the search
    # (as the timed step calls it: an iterative fit -- the form the committed PMC pass counted)
    t_v = time_kernel(lambda: model.v2v_min(verts, iterative=True), 10)
    alg_flop = FLOP_PER_V2V_PAIR * batch * v * v
    ach_v = alg_flop / t_v / 1e12
    ref_layout_bytes = batch * (12 * v + v * v + 8 * v)           # SURVEY.md 8(d) layout (i)
    compact_bytes = batch * (12 * v + 8 * v) + v * v // 8         # layout (ii): bit-packed mask read once
    prof = profile_constants()
    prof_v = prof['search']
    # what the vector units EXECUTE: every vector instruction of the launch (SQ_INSTS_VALU of the committed PMC pass, same
    # code, same batch) counted as a 64-lane FMA = an upper bound of the executed FLOP; `frac` is that over the FP32 vector
    # peak and can never exceed valu_busy.  The all-pairs figure of SURVEY 8(d) is `equivalent_*`: most pairs are pruned.
    exe_v = (prof_v['valu_instr'] * 64 * 2 / t_v / 1e12) if (prof_v['valu_instr'] and batch == BATCH_PER_GPU) else None
    roof = {'kernel': 'v2v_scan_kernel<pairs> (+ v2v_rows_seed, v2v_tree_finalize; beside the inside test: '
                      'v2v_scan_shared_kernel, the same code capped at 7 wavefronts per SIMD)', 'bound': 'valu',
            'achieved': round(exe_v, 2) if exe_v is not None else None, 'peak': PEAK_FP32_VECTOR_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(exe_v / PEAK_FP32_VECTOR_TFLOPS, 4) if exe_v is not None else None,
            'frac_formula': 'SQ_INSTS_VALU per launch (profiles/) x 64 lanes x 2 / launch time measured here / peak: EXECUTED '
                            'vector work with every instruction priced as an FMA (an upper bound; <= valu_busy)',
            'equivalent_achieved': round(ach_v, 2), 'equivalent_frac': round(ach_v / PEAK_FP32_VECTOR_TFLOPS, 4),
            'traffic': prof_v['traffic_bytes'] if batch == BATCH_PER_GPU else None,
            'valu_busy': prof_v['valu_busy'], 'profile_source': prof_v['source'],
            'live_lane_fraction': live_lane_fraction()[0], 'live_lane_fraction_source': live_lane_fraction()[1],
            # frac mixes an instruction count from profiles/ with a launch time measured here: stale = the kernel source has
            # changed since that count was collected (then frac / valu_busy / traffic describe OLDER code)
            'stale': prof_v['stale'], 'stale_reason': prof_v['stale_reason'],
            'launch_ms': round(t_v * 1e3, 4),
            'algorithmic_flop_per_launch': alg_flop, 'flop_per_pair': FLOP_PER_V2V_PAIR,
            'algorithmic_bytes_per_launch': compact_bytes,
            'compact_layout_GBs': round(compact_bytes / t_v / 1e9, 1),
            'reference_layout_equivalent_GBs': round(ref_layout_bytes / t_v / 1e9, 1),
            'reference_layout_equivalent_frac_of_hbm': round(ref_layout_bytes / t_v / 1e9 / PEAK_HBM_GBS, 3),
            'valu_instr_per_launch': prof_v['valu_instr'], 'scalar_instr_per_launch': prof_v['salu_instr'],
            'note': 'frac / achieved = executed vector work (see frac_formula).  equivalent_* = 8 FLOP x V^2 x B / launch time '
                    '(SURVEY 8d), an all-pairs-EQUIVALENT figure: most (column, row) pairs are never evaluated -- pruned by box '
                    'distance, by the lanes the mask leaves a row for below a node, and in groups of four rows no reachable '
                    'lane may use: pruning, not utilisation, is the win.  The mask is bit-packed and L2-resident: the '
                    'equivalent-bandwidth figures are NOT physical bandwidth'}
    # ---- the inside test: sheared strips + leaf slabs + near-leaf lists + tiles + ray_leaf_kernel + fan finalize
    t_w = time_kernel(lambda: model.exterior_flags(verts, apply_segments=False), 10)
    work = model.ray_work(verts)
    steps = work['elements']                                     # wavefront element steps, 64 rays each
    ops = OPS_PER_RAY_ELEMENT * work['queries_per_step'] * steps
    ach = ops / t_w / 1e12
    lane_instr = VALU_INSTR_PER_RAY_ELEMENT * work['queries_per_step'] * steps / t_w / 1e12
    ref_flops = FLOP_PER_WINDING_PAIR * batch * v * f
    tree = model.winding_tree_work(verts)
    tree_steps = tree['leaf_elements'] + tree['cap_elements']
    prof_r = prof['ray_leaf_kernel']
    inside = {'kernel': 'ray_leaf_kernel (+ ray_leaf_bounds, ray_near, ray_tiles_fill, ray_finalize_verts)',
              'bound': 'valu', 'achieved': round(ach, 2), 'peak': PEAK_FP32_VECTOR_TFLOPS, 'unit': 'TFLOP/s',
              'frac': round(ach / PEAK_FP32_VECTOR_TFLOPS, 4),
              'frac_formula': '21 executed FP32 operations x 64 rays x element steps (counted by tuch_ray_work on this input) '
                              '/ launch-group time measured here / peak: EXECUTED work',
              'equivalent_frac': round(ref_flops / t_w / 1e12 / PEAK_FP32_VECTOR_TFLOPS, 2),
              'plain_issue_T_lane_instr_per_s': round(lane_instr, 2), 'plain_issue_peak': PEAK_PLAIN_ISSUE_TLANEOPS,
              'frac_of_plain_issue_peak': round(lane_instr / PEAK_PLAIN_ISSUE_TLANEOPS, 4),
              'traffic': prof_r['traffic_bytes'] if batch == BATCH_PER_GPU else None,
              'valu_busy': prof_r['valu_busy'], 'profile_source': prof_r['source'],
              'stale': prof_r['stale'], 'stale_reason': prof_r['stale_reason'],
              'launch_ms': round(t_w * 1e3, 4),
              'executed_op_per_launch': ops, 'op_per_ray_element': OPS_PER_RAY_ELEMENT,
              'element_steps_per_launch': steps, 'wavefront_tiles': work['wavefronts'],
              'rays_inside_leaf_slabs_when_block_major': round(work['lanes_inside_leaf_slabs'], 3),
              # the same vertices through last round's kernel (exact solid angles over the cluster tree), for scale
              'solid_angle_tree_walk_steps': tree_steps, 'solid_angle_tree_walk_flop': FLOP_PER_STRIP_ELEMENT * 64 * tree_steps,
              # the reference's formulation (every query x every face) priced at this launch time: far
              # above the vector peak because crossings are counted, not solid angles summed
              'reference_formulation_flop_per_launch': ref_flops,
              'reference_formulation_equivalent_TFLOPs': round(ref_flops / t_w / 1e12, 1),
              'algorithmic_bytes_per_launch': batch * (v * 12 + v) + f * 12,
              'note': 'the whole launch group is timed; ray_leaf_kernel alone is ~0.5 of it (98 of ~200 us) and keeps the vector '
                      'units ~0.75 busy (valu_busy): its two edge-function products are one packed multiplication, the '
                      'rest is plain FP32 / integer work the packed forms do not cover (min3 / max3, compares)'}
    return roof, inside, verts, model


def contact_loss_eval(p, batch, verts, model):
    """BASELINE metric 2, 'contact-loss eval ms/body' (SURVEY.md §8d): RegressorLoss.contact_loss
    (tuch/train/loss.py:240-317) on the posed vertices, batch 64, forward and forward+backward,
    plain and HD branch; HIP events on the launch stream."""
    dev = verts.device
    valid = torch.ones(batch, dtype=torch.bool, device=dev)
    out = {}
    # TUCH.contact_from_verts (tuch/train/train_module.py:69-91): minimum squared distance of ALL region pairs,
    # unmasked; the reference calls it twice per training step
    if model.num_pairs:
        with torch.no_grad():
            out['contact_from_verts_ms'] = round(time_kernel(
                lambda: model.region_pair_min(verts, select=None, masked=False), 5) * 1e3, 4)
        out['contact_from_verts_pairs'] = model.num_pairs
    # FRESH bodies: a training loop calls the loss on bodies it has never seen (tuch/train/train_module.py:302-317), while
    # timing one tensor over and over hands the nearest-vertex search its own previous answer as the seed.  *_fresh_*: K
    # distinct pose batches in turn -- the seed of every call is ANOTHER batch's answer; the captured step gets its inputs
    # rewritten in place between replays.  The figures without the suffix are the same-bodies (hinted) ones of rounds 1-4.
    probs = fresh_problems(p, dev, 1002)
    with torch.no_grad():
        fresh_verts = [verts] + [q['smpl'](global_orient=q['global_orient'], body_pose=q['body_pose'],
                                           betas=q['betas']).vertices for q in probs[1:]]
    out['fresh_batches'] = len(fresh_verts)
    for tag, use_hd in (('plain', False), ('hd', True)):
        crit = regressor_loss(p, use_hd)
        v = verts.clone().requires_grad_(True)
        turn = {'i': 0}

        def next_verts():
            turn['i'] = (turn['i'] + 1) % len(fresh_verts)
            with torch.no_grad():
                v.copy_(fresh_verts[turn['i']])

        def fwd():
            with torch.no_grad():
                return crit.contact_loss(v, valid)

        def fwd_bwd():
            v.grad = None
            crit.contact_loss(v, valid).backward()
        iters = 5 if not use_hd else 3
        out['regressor_%s_fwd_ms_per_body' % tag] = round(time_kernel(fwd, iters) * 1e3 / batch, 5)
        out['regressor_%s_fwd_bwd_ms_per_body' % tag] = round(time_kernel(fwd_bwd, iters) * 1e3 / batch, 5)
        out['regressor_%s_fwd_fresh_ms_per_body' % tag] = round(time_kernel(fwd, 2 * iters, next_verts) * 1e3 / batch, 5)
        out['regressor_%s_fwd_bwd_fresh_ms_per_body' % tag] = round(time_kernel(fwd_bwd, 2 * iters, next_verts) * 1e3 / batch, 5)
        with torch.no_grad():
            v.copy_(verts)
        out['train_style_%s_step_ms' % tag] = round(time_kernel(make_train_step(p, use_hd), iters) * 1e3, 4)
        # the same step captured once and replayed as one hipGraph (what a training loop that captures its step pays:
        # the eager figure above includes the host's launch rate, ~50 launches per step)
        try:
            replay = capture(make_train_step(p, use_hd, fresh=probs), 3)
            out['train_style_%s_step_graph_ms' % tag] = round(time_kernel(replay, 10) * 1e3, 4)
            out['train_style_%s_step_graph_fresh_ms' % tag] = round(time_kernel(replay, 12, replay.next_bodies) * 1e3, 4)
        except Exception as e:                                   # noqa: BLE001 -- reported, not fatal for the bench line
            out['train_style_%s_step_graph_ms' % tag] = 'capture failed: %s' % type(e).__name__
        if use_hd:
            # the cost of bit-exact index work in the HD branch: option hd_search=0 is the exact VALU search (v2v_indexed_kernel:
            # partners identical to a float32 brute force, first index on ties); the default search on the matrix cores is exact
            # up to ties within 2e-6 relative of the squared distance (3 of 354 048 picks at batch 64, each verified per point in
            # tests/test_gpu_properties.py) -- the reference's own picks are decided by ~4e-6 of bmm rounding there
            crit._model.set_option('hd_search', 0)
            try:
                out['hd_exact_mode_fwd_bwd_ms_per_body'] = round(time_kernel(fwd_bwd, iters) * 1e3 / batch, 5)
                replay = capture(make_train_step(p, use_hd, fresh=probs), 3)
                out['hd_exact_mode_step_graph_ms'] = round(time_kernel(replay, 10) * 1e3, 4)
            except Exception as e:                               # noqa: BLE001
                out['hd_exact_mode_step_graph_ms'] = 'failed: %s' % type(e).__name__
            finally:
                crit._model.set_option('hd_search', 1)
    out['fresh_note'] = ('*_fresh_*: %d distinct pose batches rotated through the same call / the same captured step (inputs '
                         'written in place between replays, those small copies included): the nearest-vertex search is '
                         'seeded by ANOTHER batch\'s partners, as in a training loop; without the suffix: identical bodies '
                         'every call (the search is seeded by its own previous answer, as in an iterative fit)' % len(fresh_verts))
    return out


def shard_sweep(device, seed):
    """The per-GPU shards of a global batch of 64 (64 / 32 / 16 / 8 bodies per GPU at 1 / 2 / 4 / 8
    GPUs), each timed on this one GPU: the stage-2 step replayed as a hipGraph and launched eagerly."""
    out = {}
    for b in (8, 16, 32, 64):
        p = build_problem(b, device, seed)
        eager = time_kernel(make_step(p), 10) * 1e3
        graph = time_kernel(capture(make_step(p), 3), 20) * 1e3
        out[str(b)] = {'graph_ms': round(graph, 4), 'eager_ms': round(eager, 4),
                       'body_iterations_per_s_graph': round(b / graph * 1e3, 1),
                       'implied_8gpu_global64_body_iterations_per_s': round(64 / graph * 1e3, 1) if b == 8 else None}
    return out


def _checked(fn):
    try:
        return fn()
    except Exception as exc:                     # noqa: BLE001 -- the measurement stands; the check reports what went wrong
        return {'ok': False, 'error': repr(exc)}


def workloads(device, seed):
    """Per-rank workloads of BASELINE configs 3, 4, 5 on this GPU (synthetic rotation matrices stand in for the
    frozen regressor's output)."""
    out = {}
    p32 = build_problem(32, device, seed + 1)
    fit, fitter = make_fit(p32, 100)
    t = time_kernel(fit, 1)
    out['config3_fit_b32_100+100_iters'] = {'seconds_per_fit': round(t, 4), 'body_iterations_per_s': round(32 * 200 / t, 1),
                                            'graph_replayed': dict(fitter.graph_replayed)}
    tuch_step = make_tuch_step(p32, run_smplify=False)
    eager_ms = time_kernel(tuch_step, 5) * 1e3
    graph_fresh_ms = None
    try:        # our part of the step has no host synchronisation: the whole step replays as one hipGraph
        replay4 = capture(tuch_step, 3)
        graph_ms = round(time_kernel(replay4, 10) * 1e3, 4)
        graph_fresh_ms = round(time_kernel(replay4, 12, replay4.next_bodies) * 1e3, 4)
    except RuntimeError as e:
        graph_ms = 'capture failed: %s' % str(e).splitlines()[0]
    out['config4_shard_b32_train_step'] = {
        'ms': round(eager_ms, 4), 'graph_ms': graph_ms, 'graph_fresh_ms': graph_fresh_ms,
        'fresh_ms': round(time_kernel(tuch_step, 8, tuch_step.next_bodies) * 1e3, 4),
        'selfcheck': _checked(lambda: check_tuch_step(p32, tuch_step, False)),
        'contact_only_plain_ms': round(time_kernel(make_train_step(p32, False), 5) * 1e3, 4),
        'contact_only_hd_ms': round(time_kernel(make_train_step(p32, True), 3) * 1e3, 4),
        'what': 'TUCH.forward_train_step (no SMPLify in the loop) + backward, 32 bodies per rank (256 / 8), stand-in '
                'regressors; ms = eager (CPU-launch-bound: ~580 small launches), graph_ms = the same step captured once and '
                'replayed; contact_only_* = SMPL fwd (pose2rot=False) + RegressorLoss.contact_loss + backward alone; '
                '*fresh_ms = %d different input batches in turn, written in place between calls / replays (the bodies are new '
                'every step, as in training; ms / graph_ms repeat ONE batch, whose searches are seeded by their own previous '
                'answer)' % FRESH_BATCHES}
    p64 = build_problem(64, device, seed + 2)
    step5 = make_tuch_step(p64, run_smplify=True, smplify_iters=10)
    ms5 = round(time_kernel(step5, 2) * 1e3, 4)
    fresh5 = round(time_kernel(step5, 4, step5.next_bodies) * 1e3, 4)
    graph_fresh5 = None
    try:        # the whole step as ONE hipGraph: the SMPLify-DC iterations are unrolled into the enclosing capture
        replay5 = capture(step5, 3)
        graph5 = round(time_kernel(replay5, 5) * 1e3, 4)
        graph_fresh5 = round(time_kernel(replay5, 8, replay5.next_bodies) * 1e3, 4)
    except Exception as e:                                       # noqa: BLE001 -- reported, not fatal for the bench line
        graph5 = 'capture failed: %s: %s' % (type(e).__name__, str(e).splitlines()[0][:200])
    out['config5_shard_b64_in_the_loop_step'] = {
        'ms': ms5, 'graph_ms': graph5, 'fresh_ms': fresh5, 'graph_fresh_ms': graph_fresh5,
        'selfcheck': _checked(lambda: check_tuch_step(p64, step5, True)),
        'what': 'TUCH.forward_train_step with --run_smplify (SMPLify-DC 10 + 10 iterations with contact in the loop) + '
                'backward, 64 bodies per rank (512 / 8); ms = eager step (its SMPLify loops replay their own kept graphs), '
                'graph_ms = the whole step captured once and replayed as one hipGraph (the loops unrolled into it); the '
                'bf16 ResNet regressor is stock PyTorch and not part of the path; *fresh_ms as for config 4 (the fits of the '
                'SMPLify-DC loop inside are iterative: from their second iteration on the search is seeded by the previous '
                'iteration whatever the bodies)'}
    return out


def kernels_per_step(step):
    """Device kernels one call of `step` launches (for a captured step: the kernel nodes of one replay), counted with the
    torch profiler's device-side records; memory copies / fills are reported separately."""
    try:
        from torch.profiler import ProfilerActivity, profile
        step()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):          # the tracer now and then drops device records of a graph replay: it can only under-count
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                step()
                torch.cuda.synchronize()
            names = [e.name for e in prof.events() if str(getattr(e, 'device_type', '')).endswith('CUDA')]
            # (torch's zeros_/fill_ run as an elementwise kernel with a FillFunctor: a fill, whatever engine executes it)
            copies = [n for n in names if 'memcpy' in n.lower() or 'memset' in n.lower() or 'copyBuffer' in n or 'fillBuffer' in n
                      or 'FillFunctor' in n]
            got = {'kernels': len(names) - len(copies), 'copies_and_fills': len(copies)}
            if best is None or got['kernels'] + got['copies_and_fills'] > best['kernels'] + best['copies_and_fills']:
                best = got
        return best
    except Exception as exc:
        return {'error': repr(exc)}


def worst_case(device, seed, batch, folded=False):
    """All bodies self-penetrating (arm across the torso, legs together) instead of half of them; folded: a third with a
    forearm THROUGH the trunk, a third with the legs crossed through each other, a third folded over the thighs
    (synthetic.folded_poses) -- where the near-leaf lists and the pair list grow."""
    from tuch_amd.smplify.losses import contact_model_for
    p = build_problem(batch, device, seed, penetrating_fraction=1.0, folded=folded)
    ms = time_kernel(capture(make_step(p), 3), 20) * 1e3
    model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
    with torch.no_grad():
        verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices
        work = model.winding_tree_work(verts)
        interior = float((model.exterior_flags(verts, apply_segments=True) == 0).float().sum(1).mean())
    return {'ms_per_step': round(ms, 4), 'body_iterations_per_s': round(batch / ms * 1e3, 1),
            'ray_element_steps': model.ray_work(verts)['elements'],
            'solid_angle_tree_walk_steps': work['leaf_elements'] + work['cap_elements'],
            'mean_interior_vertices_per_body': round(interior, 1),
            'what': 'batch %d, penetrating_fraction=1.0 (default mix: 0.5)' % batch}


def irregular_topology(device, seed, batch):
    """The same step on the irregular-topology body (V=6762: geodesic icosahedron with random edge flips, valence 4-9,
    painted segments with ragged boundaries) next to the lat-long sphere's numbers: what the pruning statistics look like
    on a mesh that is not a regular grid."""
    from tuch_amd.smplify.losses import contact_model_for
    out = {}
    for topo in ('uv', 'ico'):
        p = build_problem(batch, device, seed, topology=topo)
        ms = time_kernel(capture(make_step(p), 3), 20) * 1e3
        model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
        with torch.no_grad():
            verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices
            interior = float((model.exterior_flags(verts, apply_segments=True) == 0).float().sum(1).mean())
            t_search = time_kernel(lambda: model.v2v_min(verts), 10) * 1e3
            t_inside = time_kernel(lambda: model.exterior_flags(verts, apply_segments=True), 10) * 1e3
        body = p['body']
        valence = np.bincount(np.bincount(body.faces.ravel()))
        tree = ops_cluster_tree_info(model)
        out[topo] = {'V': body.num_verts, 'F': body.num_faces, 'valence_min_max': [int(np.nonzero(valence)[0][0]), len(valence) - 1],
                     'segments': len(body.segments), 'ms_per_step': round(ms, 4),
                     'ray_element_steps': model.ray_work(verts)['elements'], 'search_ms': round(t_search, 4),
                     'inside_test_with_segments_ms': round(t_inside, 4), 'mean_interior_vertices_per_body': round(interior, 1),
                     'strip_stream_elements': tree['exact_len'], 'leaves': tree['leaves']}
    return out


def ops_cluster_tree_info(model):
    """(leaf-strip stream length, number of leaves) of the model's cluster tree, through the host-side builder."""
    from tuch_amd import ops
    t = ops.cluster_tree(model.faces_np, model.num_verts, leaf_faces=max(model.num_faces // 850, 32))
    return {'exact_len': int(t['exact_len']), 'leaves': int((t['nodes'][:, 3] > 0).sum())}


def headline_without_hints(device, seed, batch):
    """The headline step with the search's partner hints switched OFF (option v2v_hint = 0): every iteration's search starts
    from the hint-free seed, as the first call on new bodies does.  (An iterative fit is what the hints are for -- Adam
    moves a pose by 1e-2 per iteration --; this line says what they are worth.)"""
    from tuch_amd.smplify.losses import contact_model_for
    p = build_problem(batch, device, seed)
    model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
    model.set_option('v2v_hint', 0)
    try:
        ms = time_kernel(capture(make_step(p), 3), 20) * 1e3
    finally:
        model.set_option('v2v_hint', 1)
    return {'ms_per_step': round(ms, 4), 'body_iterations_per_s': round(batch / ms * 1e3, 1)}


def float_atomics_cost(device, seed, batch):
    """The step with TUCH_DETERMINISTIC=0 (gradient scatters through float atomics: last-ulp run-to-run noise) -- the
    headline runs in the default, deterministic mode (64-bit fixed-point integer atomics: bit-reproducible fits);
    captured and replayed like the headline."""
    from tuch_amd import ops
    p = build_problem(batch, device, seed)
    with ops.deterministic_mode(False):
        ms = time_kernel(capture(make_step(p), 3), 20) * 1e3
    # the default mode timed the SAME way (a fresh capture, best of two 20-replay passes): the headline's block timing is not
    # comparable with a best-of-two figure
    with ops.deterministic_mode(True):
        det = time_kernel(capture(make_step(p), 3), 20) * 1e3
    return {'ms_per_step': round(ms, 4), 'body_iterations_per_s': round(batch / ms * 1e3, 1),
            'deterministic_same_protocol_ms': round(det, 4)}


def cpu_baseline(p, seconds):
    """The CPU oracle (a port of the reference's arithmetic, test infrastructure) timed on the host:
    contact loss forward + gradient for whole bodies, until ~`seconds` have elapsed."""
    from oracle import contact as oc
    body = p['body']
    with torch.no_grad():
        verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'],
                          betas=p['betas']).vertices.cpu().numpy()
    gm = body.geodesics > 0.3
    segs = [oc.Segment(n, body.faces, s['vidx'], list(s['bands'].values())) for n, s in body.segments.items()]
    cores = oc.max_threads()
    oc.smplify_contact_body(verts[0], body.faces, gm, 0.02, segs, None)      # warm-up
    t0, n = time.time(), 0
    while time.time() - t0 < seconds and n < verts.shape[0]:
        oc.smplify_contact_body(verts[n], body.faces, gm, 0.02, segs, None)
        n += 1
    dt = time.time() - t0
    return {'value': round(n / dt, 3), 'unit': 'body-fit iterations/s', 'cores': cores, 'kind': 'port',
            'sample': '%d bodies (V=6890,F=13776): contact term forward+gradient (winding, segments, '
                      'masked v2v, push/pull) with the C/OpenMP oracle, %.1f s; SMPL forward, '
                      'reprojection and Adam are excluded (negligible on CPU)' % (n, dt)}


def selfcheck(p, step, sample=(0, 37)):
    """After the timed blocks: (1) the objective the REPLAYED step reports equals an eager evaluation (separate launches of
    the same HIP path) at the same parameters -- the graph does the work it was captured for, on the current state; (2)
    at that state, two sampled bodies against the CPU oracle (the checker, tests/ infrastructure -- not the thing
    measured): posed vertices vs the oracle's LBS, contact value of losses.py:96-105 vs oracle/contact.py."""
    from oracle import contact as oc
    from oracle import lbs as ol
    from oracle import smplify as osm
    from tuch_amd.ops import MODE_SMPLIFY, contact_terms
    from tuch_amd.smplify.losses import contact_model_for
    eager, verts, joints, pose = step.objective()
    replayed = float(step()[0])                      # reports the loss at the parameters it started from, then updates
    torch.cuda.synchronize()
    body = p['body']
    gm = body.geodesics > 0.3
    segs = [oc.Segment(n, body.faces, s['vidx'], list(s['bands'].values())) for n, s in body.segments.items()]
    model = contact_model_for(p['geomask'], p['face_tensor'], p['segments'], p['cdict'])
    ext, _, partner, _ = model.exterior_and_partner(verts, apply_segments=True)
    per_body, _ = contact_terms(verts, partner, ext, None, MODE_SMPLIFY, 0.02)
    m = ol.model_tensors(body)
    worst_c = worst_v = worst_d = 0.0
    flag_diff_clear = flag_diff_near = partner_diff = 0
    sample = [b for b in sample if b < verts.shape[0]]
    for b in sample:
        vb = verts[b].cpu().numpy()
        r = oc.smplify_contact_body(vb, body.faces, gm, 0.02, segs, None)
        ext_gpu = ext[b].cpu().numpy().astype(bool)
        part_gpu = partner[b].cpu().numpy().astype(np.int64)
        # flags: identical unless the winding number sits within 1e-4 of the threshold (or the vertex touches a triangle:
        # not checked here, the tests do); partners: identical up to ties within the reference's own distance noise
        differ = ext_gpu != r['exterior']
        near = np.abs(r['winding'] - 0.99) < 1e-4
        flag_diff_near += int((differ & near).sum())
        flag_diff_clear += int((differ & ~near).sum())
        v64 = vb.astype(np.float64)
        d_gpu = ((v64 - v64[part_gpu]) ** 2).sum(1)
        d_ref = ((v64 - v64[r['argmin']]) ** 2).sum(1)
        partner_diff += int((part_gpu != r['argmin']).sum())
        worst_d = max(worst_d, float(np.abs(d_gpu - d_ref).max()))
        # the value, by the oracle's arithmetic on the device's own flags and partners (losses.py:96-105)
        _, dist = oc._pair_distance(vb, part_gpu)
        inside, _ = oc._tanh2_terms(dist, ~ext_gpu, 1.0, 0.04)
        outside, _ = oc._tanh2_terms(dist, ext_gpu & (dist < np.float32(0.02)), 0.005, 0.005)
        want = inside + outside
        worst_c = max(worst_c, abs(float(per_body[b]) - want) / max(abs(want), 1e-6))
        # (global_orient moves too; the vertices of the oracle's LBS at the same pose are compared up to that rotation
        # through pairwise distances of a vertex subset, which a rigid motion leaves unchanged)
        ov, _ = ol.smpl_forward(m, p['betas'][b:b + 1].cpu(), pose[b:b + 1].cpu(), torch.zeros(1, 3))
        idx = np.arange(0, vb.shape[0], 97)
        d_g = np.linalg.norm(vb[idx][:, None] - vb[idx][None], axis=2)
        ovn = ov[0].numpy()
        d_r = np.linalg.norm(ovn[idx][:, None] - ovn[idx][None], axis=2)
        worst_v = max(worst_v, float(np.abs(d_g - d_r).max()))
    graph_err = abs(replayed - eager) / max(abs(eager), 1e-12)
    return {'graph_vs_eager_rel_err': graph_err, 'objective': eager, 'oracle_bodies': sample,
            'contact_value_max_rel_err': worst_c, 'lbs_pairwise_distance_max_abs_err_m': worst_v,
            'exterior_flags_differ_clear': flag_diff_clear, 'exterior_flags_differ_within_1e-4_of_threshold': flag_diff_near,
            'partners_differ': partner_diff, 'partner_d2_max_abs_diff': worst_d,
            'max_rel_err': max(worst_c, graph_err),
            'ok': bool(worst_c < 1e-4 and worst_v < 1e-4 and graph_err <= 1e-4 and flag_diff_clear == 0 and worst_d < 2e-6)}


def rccl_smoke_child():
    """Runs in a child process (`bench.py --rccl-smoke`): RCCL on the one GPU there is.  A process group of ONE rank
    over backend nccl; a device-tensor all-reduce; RegressorLoss(global_mean=True).contact_loss -- whose count of valid
    bodies is all-reduced on the calling stream (tuch_amd/dist.py) -- captured in a hipGraph with the group alive and
    replayed; destroy_process_group.  Prints one JSON line."""
    import torch.distributed as dist
    from tuch_amd import dist as tdist
    out = {'backend': 'nccl (RCCL)', 'world_size': 1}
    device = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    torch.cuda.set_stream(torch.cuda.Stream(device=device))
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % free_port(), rank=0, world_size=1, device_id=device)
    try:
        tdist.REDUCE_SINGLE_RANK = True
        t = torch.tensor([3.5, 64.0], device=device)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        out['all_reduce'] = t.tolist()
        total, count = tdist.allreduce_loss(torch.tensor(2.25, device=device), 64)
        out['allreduce_loss'] = [float(total), float(count)]
        p = build_problem(8, device, seed=1002)
        import types
        from tuch_amd.train.loss import RegressorLoss
        body = p['body']
        crit = RegressorLoss(types.SimpleNamespace(contact_loss_weight=1.0), device, body.num_verts, p['face_tensor'],
                             torch.tensor(body.geodesics, device=device), geothres=0.3, euclthres=0.02,
                             face_tensor=p['face_tensor'], use_hd=False, segments=p['segments'], global_mean=True)
        with torch.no_grad():
            verts = p['smpl'](global_orient=p['global_orient'], body_pose=p['body_pose'], betas=p['betas']).vertices
        v = verts.clone().requires_grad_(True)
        valid = torch.ones(8, dtype=torch.bool, device=device)
        valid[3] = False

        def fwd_bwd():
            v.grad = None
            loss = crit.contact_loss(v, valid)
            loss.backward()
            return loss
        eager = float(fwd_bwd())
        grad_eager = v.grad.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                fwd_bwd()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        v.grad = None
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):
            captured = fwd_bwd()
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        out['captured_contact_loss'] = {'eager': eager, 'replayed': float(captured),
                                        # (float atomics: the order of the additions differs from run to run)
                                        'grad_equal': bool((v.grad - grad_eager).abs().max() <= 2e-6 * grad_eager.abs().max()),
                                        'grad_max_abs_diff': float((v.grad - grad_eager).abs().max()),
                                        'grad_max_abs': float(grad_eager.abs().max())}
        ok = out['all_reduce'] == [3.5, 64.0] and abs(float(captured) - eager) <= 1e-6 * abs(eager) \
            and out['captured_contact_loss']['grad_equal']
        out['status'] = 'ok' if ok else 'mismatch'
    finally:
        tdist.REDUCE_SINGLE_RANK = False
        dist.destroy_process_group()
    out['destroyed'] = True
    print(json.dumps(out), flush=True)


def rccl_smoke():
    """rccl_smoke_child in a process of its own, under a timeout: a wedged collective must not take the bench line with it."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    try:
        res = subprocess.run([sys.executable, os.path.abspath(__file__), '--rccl-smoke'], env=env, capture_output=True, text=True,
                             timeout=240)
    except subprocess.TimeoutExpired:
        return {'status': 'timeout after 240 s'}
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    if res.returncode != 0 or not lines:
        return {'status': 'failed (rc %d)' % res.returncode, 'stderr_tail': res.stderr[-600:]}
    out = json.loads(lines[-1])
    out['note'] = 'one rank on the one GPU: exercises RCCL init, a device all-reduce, capture + replay of a collective, ' \
                  'destroy; no N > 1 value exists until the driver runs this script on a multi-GPU node'
    return out


def cpu_torch_chain(p):
    """One body through the reference's stock-op formulation on CPU (materialises the 3.4 GB
    [1,Q,F,3,3] tensor like tuch/utils/contact.py:79): first call and warm call, all host threads."""
    import psutil
    from oracle import torch_chain as tc
    if psutil.virtual_memory().available < 24 * 2 ** 30:
        return {'skipped': 'less than 24 GiB of free host memory'}
    body = p['body']
    with torch.no_grad():
        verts = p['smpl'](global_orient=p['global_orient'][:1], body_pose=p['body_pose'][:1],
                          betas=p['betas'][:1]).vertices[0].cpu()
        faces = torch.tensor(body.faces)
        gm = torch.tensor(body.geodesics > 0.3)
        times = []
        for _ in range(2):
            t0 = time.time()
            tc.contact_forward_one_body(verts, faces, gm, 0.02)
            times.append(time.time() - t0)
    return {'value': round(1.0 / times[1], 4), 'unit': 'body contact-loss forwards/s', 'cores': torch.get_num_threads(),
            'kind': 'port', 'first_call_s': round(times[0], 2), 'warm_call_s': round(times[1], 2),
            'sample': '1 body (V=6890,F=13776), contact forward only (pairwise + winding + argmin + terms), '
                      'torch CPU ops materialising the reference intermediates; no segments, no backward'}


CONFIGS = {
    '2': dict(metric='SMPLify-DC fit iters/sec at batch 64', unit='body-fit iterations/s', iters_per_step=1,
              workload='configs[1] extended to the full stage-2 step: batch=%d/GPU SMPL forward + contact_fitting_loss '
                       '(L_P/L_C push/pull, winding inside test, segment filter, r2r) + backward + Adam, V=6890 '
                       'F=13776, float32'),
    '3': dict(metric='SMPLify-DC fit iters/sec, demo-style 100+100-iteration fits', unit='body-fit iterations/s',
              iters_per_step=200,
              workload='configs[2]: demo_smplify_dc.py-style fit, batch=%d/GPU, 100 stage-1 + 100 stage-2 iterations per '
                       'step (SMPLifyDC.__call__, loops replayed as hipGraphs), V=6890 F=13776, float32'),
    '4-shard': dict(metric='train.py-style contact-loss steps/sec (bodies/s)', unit='bodies/s', iters_per_step=1,
                    workload='configs[3] per-rank shard: batch=%d/GPU, TUCH.forward_train_step (regressor stand-in -> SMPL '
                             'with rotation matrices -> fits dictionary -> RegressorLoss incl. the HD contact term, global '
                             'valid mean) + backward'),
    '5-shard': dict(metric='SMPLify-DC in-the-loop training steps/sec (bodies/s)', unit='bodies/s', iters_per_step=1,
                    workload='configs[4] per-rank shard: batch=%d/GPU, TUCH.forward_train_step with --run_smplify (SMPLify-DC '
                             '10+10 iterations with contact in the loop) + RegressorLoss (HD contact term) + backward; '
                             'regressor stand-in'),
}


def main():
    args = parse()
    if args.rccl_smoke:
        return rccl_smoke_child()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(launch_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    # one rank per GPU; TUCH_BENCH_BACKEND=gloo lets several ranks share one GPU to smoke-test the
    # multi-rank control path on a single-GPU box (RCCL refuses two ranks on one device)
    backend = os.environ.get('TUCH_BENCH_BACKEND', 'nccl')
    local = local % torch.cuda.device_count() if backend == 'gloo' else local
    device = torch.device('cuda', local)
    torch.cuda.set_device(local)
    # everything below runs on a created stream: hipGraph replays on the legacy NULL stream are not reliably ordered
    # against the work around them (tuch_amd/ops.py:off_default_stream)
    torch.cuda.set_stream(torch.cuda.Stream(device=device))
    # N > 1 measures what SURVEY 8(e) / north_star name: the GLOBAL batch of 64 split over the ranks (64 / 32 / 16 / 8 bodies
    # per GPU at 1 / 2 / 4 / 8 GPUs) = strong scaling; the weak-scaling figure (64 bodies per GPU) is timed in the same run
    # and reported as `weak_scaling`.  N = 1 is the same workload either way.
    default_strong = (world > 1 and args.global_batch is None and not args.weak and args.config == '2'
                      and args.batch == BATCH_PER_GPU and BATCH_PER_GPU % world == 0)
    if default_strong:
        args.global_batch = BATCH_PER_GPU
    if args.global_batch is not None:
        if args.global_batch % world:
            raise SystemExit('--global-batch %d is not divisible by %d ranks' % (args.global_batch, world))
        batch, scaling = args.global_batch // world, 'strong'
    else:
        batch = {'3': 32, '4-shard': 32}.get(args.config, args.batch) if args.batch == BATCH_PER_GPU else args.batch
        scaling = 'weak'
    torch.manual_seed(1000 + rank)
    p = build_problem(batch, device, seed=1002 + rank)
    launch = 'eager'
    if args.config == '2':
        step = make_step(p)
        if not args.eager:
            # the whole step replayed as one hipGraph.  Captured BEFORE the process group exists: the step holds no
            # collective, and a capture next to RCCL's watchdog thread is the one combination that faulted in round 1
            step = capture(step, args.warmup)
            launch = 'hipGraph replay of the whole step'
    elif args.config == '3':
        step = make_fit(p, 100)[0]
        launch = 'SMPLifyDC.__call__ (each loop replayed as a hipGraph after 3 eager iterations)'
    else:
        # a training loop: every step sees a NEW input batch (FRESH_BATCHES of them in turn, written in place)
        one = make_tuch_step(p, run_smplify=args.config == '5-shard', smplify_iters=10)
        step = lambda: (one.next_bodies(), one())[1]
        launch = 'eager; %d different input batches in turn' % FRESH_BATCHES
    weak_step = None
    if default_strong and not args.eager:
        # the weak-scaling twin (64 bodies on every rank), captured before the process group exists like the headline
        weak_step = capture(make_step(build_problem(BATCH_PER_GPU, device, seed=2002 + rank)), args.warmup)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            torch.distributed.init_process_group('nccl', device_id=device)
        else:
            torch.distributed.init_process_group(backend)

    def reduce(local_stats):
        fresh = isinstance(local_stats, tuple)
        if fresh:                                        # (loss sum, bodies) as two 0-d tensors -> one new [2] tensor
            local_stats = torch.stack([t.to(torch.float32).reshape(()) for t in local_stats])
        if os.environ.get('TUCH_BENCH_DEBUG'):
            print('rank', rank, 'local stats', local_stats.tolist(), flush=True)
        if world > 1:
            if backend == 'gloo':                        # single-GPU smoke test of the N>1 path only
                host = local_stats.cpu()
                torch.distributed.all_reduce(host)
                return host.to(local_stats.device)
            if not fresh:                                # (never the step's own output buffer: a replayed graph rewrites it)
                local_stats = local_stats.clone()
            torch.distributed.all_reduce(local_stats)    # 2 floats over RCCL / xGMI
        return local_stats

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    per_step = world > 1 and not args.allreduce_per_block

    def block(steps, step=step, per_step=per_step):
        """EXACTLY `steps` steps between two fences; seconds, MAX over ranks."""
        fence()
        t0 = time.perf_counter()
        # the fits of different bodies never exchange data (SMPLify-DC optimises every body on its own): no gradient or
        # parameter crosses ranks.  What SURVEY 8(e) / north_star specify is ONE all-reduce of the two floats of statistics
        # (loss sum, bodies) PER STEP -- the reported scalar of every iteration; it is enqueued behind the step on the
        # device (RCCL: no host synchronisation), inside the timed loop.  --allreduce-per-block reduces once per block.
        for _ in range(steps):
            stats = step()
            if per_step:
                stats = reduce(stats)
        if not per_step:
            stats = reduce(stats)
        fence()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        if world > 1:
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        return float(tmax.item()), stats

    last = None
    for _ in range(args.warmup):
        last = step()
    if last is not None:
        reduce(last)                                     # the collective's own first-call cost stays out of the timing
    dt, stats = block(args.steps)                        # the contract's measurement
    repeats = [dt] + [block(args.steps)[0] for _ in range(max(args.repeats, 1) - 1)]
    other_form = None
    if world > 1:
        odt, _ = block(args.steps, per_step=not per_step)
        other_form = {'allreduce': 'per block' if per_step else 'per step', 'ms_per_step': round(odt / args.steps * 1e3, 4),
                      'value': round(batch * world * CONFIGS[args.config]['iters_per_step'] * args.steps / odt, 2)}
    weak = None
    if weak_step is not None:
        for _ in range(args.warmup):
            weak_step()
        wdt, wstats = block(args.steps, weak_step)
        weak = {'value': round(BATCH_PER_GPU * world * args.steps / wdt, 2), 'unit': CONFIGS['2']['unit'],
                'ms_per_step': round(wdt / args.steps * 1e3, 4), 'bodies_per_gpu': BATCH_PER_GPU,
                'global_batch': BATCH_PER_GPU * world, 'bodies': float(wstats[1].item()), 'scaling': 'weak',
                'note': 'the same step with 64 bodies on EVERY rank, timed like the headline (one %d-step block, fences + MAX '
                        'over ranks) in the same run' % args.steps}
    if rank == 0:
        cfg = CONFIGS[args.config]
        bodies = batch * world
        per_step_ms = [r / args.steps * 1e3 for r in repeats]
        line = {
            'metric': cfg['metric'], 'value': round(bodies * cfg['iters_per_step'] * args.steps / dt, 2),
            'unit': cfg['unit'], 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': scaling,
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': cfg['workload'] % batch,
                       'bodies_per_gpu': batch, 'global_batch': bodies, 'euclthres': 0.02,
                       'geothres': 0.3, 'launch': launch,
                       'parallelism': 'dp%d (bodies sharded; one 2-float all-reduce per %s; no gradient or parameter exchange)'
                                      % (world, 'step, inside the timed loop' if per_step else 'timed block'),
                       'batch_iterations_per_s': round(cfg['iters_per_step'] * args.steps / dt, 3),
                       'loss_sum': float(stats[0].item()), 'bodies': float(stats[1].item())},
            'repeat_ms_per_step': {'n': len(per_step_ms), 'median': round(float(np.median(per_step_ms)), 4),
                                   'min': round(min(per_step_ms), 4), 'max': round(max(per_step_ms), 4),
                                   'note': 'the same %d-step block timed %d times; ms_per_step/value are block 1'
                                           % (args.steps, len(per_step_ms))},
            'scaling_note': 'no multi-GPU node was available to the builder: N>1 values exist only when the driver '
                            'runs this script on one; --gpus N > 1 defaults to STRONG scaling at global batch 64 (SURVEY 8e) '
                            'and reports the weak-scaling figure (64 bodies per GPU) as `weak_scaling`' if world == 1 else
                            'global batch %d split over %d ranks (%s scaling)' % (bodies, world, scaling),
        }
        if weak is not None:
            line['weak_scaling'] = weak
        if other_form is not None:
            line['other_allreduce_form'] = other_form
        if args.config == '2':
            try:
                line['selfcheck'] = selfcheck(p, step)
            except Exception as exc:                     # the measurement stands; the check reports what went wrong
                line['selfcheck'] = {'ok': False, 'error': repr(exc)}
            line['kernels_per_step'] = kernels_per_step(step)
            line['graph_timeline'] = timeline_constants()
        roof, inside, verts, model = rooflines(p, batch)
        line['roofline'], line['roofline_inside_test'] = roof, inside
        if args.config == '2':
            # the whole step priced in SURVEY 8(d)'s units (67 FLOP per (query, face) pair of the body and of every closed
            # segment, 8 FLOP per ordered vertex pair) over the measured step time: a multiple of the vector PEAK, because
            # the step does not do that pair work (exact pruning, crossings instead of solid angles)
            from tuch_amd.ops import segment_faces
            body = p['body']
            seg_pairs = sum(len(sg['vidx']) * len(segment_faces(body.faces, sg['vidx'], list(sg['bands'].values()), body.num_verts))
                            for sg in body.segments.values())
            eq = (FLOP_PER_WINDING_PAIR * (body.num_verts * body.num_faces + seg_pairs)
                  + FLOP_PER_V2V_PAIR * body.num_verts ** 2) * batch
            line['step_equivalent_flop'] = eq
            line['step_equivalent_x_peak'] = round(eq / (dt / args.steps) / 1e12 / PEAK_FP32_VECTOR_TFLOPS, 2)
        line['mfma_use'] = ('SMPL dense matmuls (blend / skin adjoint, exact f32 MFMA) as north_star asks; DEPARTURE from its '
                            '"MFMA only for the SMPL matmuls": the HD branch\'s nearest-point search (hd_search_kernel, '
                            'RegressorLoss use_hd=True) also runs on the matrix cores -- 502 -> 215 us at batch 64, option '
                            'hd_search=0 is the exact VALU kernel; the headline step (this line) uses MFMA for SMPL only')
        if world == 1 and backend == 'nccl' and not args.no_rccl_smoke:
            line['rccl_smoke'] = rccl_smoke()
        if world == 1 and not args.no_extras:
            line['contact_loss_eval'] = contact_loss_eval(p, batch, verts, model)
            line['shard_sweep'] = shard_sweep(device, 1002)
            line['workloads'] = workloads(device, 1002)
            line['worst_case'] = worst_case(device, 1002, batch)
            line['worst_case']['folded'] = worst_case(device, 1002, batch, folded=True)
            from tuch_amd import ops as ops_mod
            line['deterministic'] = bool(ops_mod.deterministic())
            line['float_atomics_mode'] = float_atomics_cost(device, 1002, batch)
            line['headline_without_hints'] = headline_without_hints(device, 1002, batch)
            line['irregular_topology'] = irregular_topology(device, 1002, batch)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(p, args.cpu_seconds)
            if not args.no_torch_chain:
                line['cpu_baseline_torch_chain'] = cpu_torch_chain(p)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
