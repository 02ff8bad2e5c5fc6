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
    mean = tdist.global_mean_loss(pb, va)
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
