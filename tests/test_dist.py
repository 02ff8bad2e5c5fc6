"""CPU, world_size 2 over gloo: sharding the bodies and all-reducing [loss sum, count] gives the
single-process result.  Per-body losses come from the CPU oracle here; on the
GPU the same helpers wrap the HIP path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tuch_amd import dist as tdist


def test_shard_range_partitions_exactly():
    for total in (1, 7, 64, 65, 256):
        for world in (1, 2, 3, 8):
            spans = [tdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, per_body, valid, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    pb = tdist.shard_batch(torch.tensor(per_body), rank, world)
    va = tdist.shard_batch(torch.tensor(valid), rank, world)
    total, count = tdist.allreduce_loss((pb * va).sum(), float(va.sum()))
    mean = tdist.global_mean_loss(pb, va).reshape(1)      # this rank's share of the global mean
    dist.all_reduce(mean)
    if rank == 0:
        out.put((float(total), float(count), float(mean)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_matches_single_process():
    from helpers import golden, golden_mask, oracle_segments
    from oracle import contact as oc
    g, gm = golden('small'), golden_mask('small')
    segs = oracle_segments(g)
    verts = np.concatenate([g['verts'], g['verts'][::-1] * 1.01, g['verts'] * 0.99], 0)   # 6 bodies
    per_body = np.array([oc.train_contact_body(v, g['faces'], gm, 0.02, segs, False)['loss'] for v in verts],
                        np.float32)
    valid = np.array([1, 1, 0, 1, 1, 1], np.float32)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, per_body, valid, q)) for r in range(2)]
    for p in procs:
        p.start()
    total, count, mean = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert count == valid.sum()
    assert abs(total - float((per_body * valid).sum())) <= 1e-6 * abs(total)
    assert abs(mean - float(per_body[valid > 0].mean())) <= 1e-6 * abs(mean)


def _mean_worker(rank, world, port, per_body, valid, spans, out):
    """global_mean_loss on this rank's (uneven) span of the batch; also its gradient w.r.t. the rank's per-body losses."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = spans[rank]
    pb = torch.tensor(per_body[lo:hi], requires_grad=True)
    va = torch.tensor(valid[lo:hi])
    share = tdist.global_mean_loss(pb, va)
    share.backward()
    total = share.detach().reshape(1).clone()
    dist.all_reduce(total)
    parts = [None] * world
    dist.all_gather_object(parts, (lo, hi, pb.grad.numpy()))
    if rank == 0:
        out.put((float(total), parts))
    dist.barrier()
    dist.destroy_process_group()


def test_four_ranks_with_unequal_valid_counts_sum_to_the_reference_mean():
    """``RegressorLoss(global_mean=True)`` on 4 ranks whose shards hold 3 / 0 / 5 / 1 valid bodies of 4 / 2 / 5 / 3 (one
    rank without a single valid body): the SUM over the ranks of `local sum / global valid count` is
    ``contact_loss[valid_fit].mean()`` over the whole batch (tuch/train/loss.py:317), and every rank's gradient with
    respect to its own bodies is the single-process gradient (1 / global count where valid, 0 elsewhere)."""
    rng = np.random.default_rng(12)
    per_body = rng.random(14).astype(np.float32) + 0.1
    valid = np.array([1, 1, 0, 1,   0, 0,   1, 1, 1, 1, 1,   0, 1, 0], np.float32)
    spans = [(0, 4), (4, 6), (6, 11), (11, 14)]
    assert [int(valid[a:b].sum()) for a, b in spans] == [3, 0, 5, 1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mean_worker, args=(r, 4, port, per_body, valid, spans, q)) for r in range(4)]
    for p in procs:
        p.start()
    total, parts = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = torch.tensor(per_body, requires_grad=True)
    want = ref[torch.tensor(valid) > 0].mean()
    want.backward()
    assert abs(total - float(want)) <= 1e-6 * abs(float(want))
    for lo, hi, grad in parts:
        assert np.allclose(grad, ref.grad.numpy()[lo:hi], rtol=1e-6, atol=0)


def _hip_worker(rank, world, port, out):
    """Rank r evaluates RegressorLoss.contact_loss (HIP path) on its shard of the medium golden batch."""
    import types
    import golden_io as gio
    from helpers import golden, golden_mask
    from tuch_amd.train.loss import RegressorLoss
    from tuch_amd.utils.segmentation import BatchBodySegment
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g, gm = golden('medium'), golden_mask('medium')
    d = torch.device('cuda:0')                      # both ranks share the one GPU of the test box
    lo, hi = tdist.shard_range(g['verts'].shape[0], rank, world)
    face_tensor = torch.tensor(g['faces'], device=d)[None].repeat(hi - lo, 1, 1)
    segs = gio.unpack_segments(g)
    crit = RegressorLoss(types.SimpleNamespace(contact_loss_weight=1.0), d, g['verts'].shape[1], face_tensor,
                         torch.tensor(np.where(gm, 1.0, 0.0).astype(np.float32), device=d), geothres=0.3,
                         euclthres=float(g['euclthres']), face_tensor=face_tensor, use_hd=(rank >= 0),
                         segments=BatchBodySegment(list(segs.keys()), face_tensor[0], segs),
                         hd_regressor=(g['hd_idx'], g['hd_w']), hd_faces=g['hd_face'], global_mean=True)
    verts = torch.tensor(g['verts'][lo:hi], device=d, requires_grad=True)
    # global_mean=True: the all-reduce of the valid count inside contact_loss goes through dist.all_reduce_sum (gloo
    # takes host tensors only: a host copy; RCCL takes the device tensor)
    share = crit.contact_loss(verts, torch.tensor(g['valid_fit'][lo:hi], device=d))
    share.backward()
    total = share.detach().cpu().reshape(1)
    dist.all_reduce(total)
    grads = [None] * world
    dist.all_gather_object(grads, (lo, hi, verts.grad.cpu().numpy()))
    if rank == 0:
        out.put((float(total), grads))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_the_hip_path_reproduce_the_reference_mean():
    """Bodies sharded over 2 ranks (gloo, sharing the box's GPU): the sum of the per-rank shares of
    RegressorLoss.contact_loss is the reference's mean over all valid bodies (loss.py:317), and each rank's
    gradient is the reference's gradient for its bodies."""
    from helpers import golden
    g = golden('medium')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hip_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    total, grads = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert abs(total - float(g['train_hd_loss'])) <= 1e-4 * abs(float(g['train_hd_loss']))
    gv = g['train_hd_grad_verts']
    from helpers import grad_close
    for lo, hi, grad in grads:
        # as every single-evaluation gradient test: 1e-4 relative + a floor of 5e-6 of the largest entry (+ the quantised
        # derivative of a saturated tanh^2 term, helpers.TANH_QUANTUM); observed maxima are logged
        grad_close(grad, gv[lo:hi], 5e-6, '2 gloo ranks on the HIP path, bodies %d:%d' % (lo, hi), quantum=True)


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks():
    """`bench.py --gpus 2` without a launcher re-execs under torch.distributed.run and prints one line with
    n_gpus 2 (two gloo ranks sharing this box's GPU; RCCL needs one device per rank)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TUCH_BENCH_BACKEND='gloo')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    def run(*extra):
        res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                              '--repeats', '2', '--no-extras', '--no-cpu-baseline'] + list(extra),
                             env=env, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-3000:]
        lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1, res.stdout[-2000:]
        return json.loads(lines[0])
    line = run('--global-batch', '8')
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['config']['bodies_per_gpu'] == 4
    # one all-reduce of the two floats per STEP inside the timed loop is the default for N > 1; the per-block
    # form is timed beside it
    assert 'all-reduce per step' in line['config']['parallelism']
    assert line['other_allreduce_form']['allreduce'] == 'per block' and line['other_allreduce_form']['value'] > 0
    line_b = run('--global-batch', '8', '--allreduce-per-block')
    assert 'per timed block' in line_b['config']['parallelism'] and line_b['other_allreduce_form']['allreduce'] == 'per step'
    assert line_b['config']['bodies'] == 8.0
    assert line['config']['bodies'] == 8.0 and line['config']['launch'].startswith('hipGraph')
    assert np.isfinite(line['config']['loss_sum']) and line['value'] > 0 and 'weak_scaling' not in line
    # the DEFAULT for N > 1 is what SURVEY 8(e) specifies: the global batch of 64 split over the ranks (strong scaling);
    # the weak-scaling figure (64 bodies per rank) rides along
    line = run()
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['config']['bodies_per_gpu'] == 32
    assert line['config']['global_batch'] == 64 and line['config']['bodies'] == 64.0
    weak = line['weak_scaling']
    assert weak['scaling'] == 'weak' and weak['bodies_per_gpu'] == 64 and weak['bodies'] == 128.0 and weak['value'] > 0
    line = run('--weak')
    assert line['scaling'] == 'weak' and line['config']['bodies_per_gpu'] == 64 and 'weak_scaling' not in line


@pytest.mark.gpu
def test_rccl_on_one_gpu_and_the_bench_selfcheck():
    """The default bench line on this box's one GPU (short): `selfcheck` -- the replayed graph reports the objective an
    eager evaluation gives at the same parameters, two bodies of that state agree with the CPU oracle -- and
    `rccl_smoke` -- a one-rank process group over RCCL: init, a device all-reduce, RegressorLoss(global_mean=True)
    .contact_loss with its all-reduced count captured in a hipGraph and replayed, destroy_process_group."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'TUCH_BENCH_BACKEND'):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '5', '--warmup', '2', '--repeats', '1',
                          '--no-extras', '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    sc = line['selfcheck']
    assert sc.get('ok'), sc
    assert sc['graph_vs_eager_rel_err'] < 1e-5 and sc['contact_value_max_rel_err'] < 1e-4
    smoke = line['rccl_smoke']
    assert smoke.get('status') == 'ok' and smoke.get('destroyed'), smoke
    assert smoke['all_reduce'] == [3.5, 64.0] and smoke['captured_contact_loss']['grad_equal']
    assert line['kernels_per_step'].get('kernels', 0) > 10, line['kernels_per_step']
    assert line['roofline']['valu_busy'] is not None and line['roofline']['traffic'] is not None
