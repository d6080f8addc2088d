"""not gpu: the N>1 path on CPU -- two processes over gloo (127.0.0.1).
Checks the rank<->worker<->shard plan, the unique-id hand-off and the delta-sum / average replica
exchange protocol (the same arithmetic the library runs over RCCL on GPUs)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from word2bits_amd import replicas


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # unique id: created on rank 0 only, identical everywhere afterwards
        uid = replicas.exchange_unique_id(dist, rank, lambda: bytes(range(128)))
        assert uid == bytes(range(128))
        # plan: contiguous worker blocks, disjoint, covering
        off, per = replicas.worker_plan(8, world, rank)
        starts = replicas.token_shard_starts(1000, 8, off, per)
        gathered = [None] * world
        dist.all_gather_object(gathered, (off, per, starts.tolist()))
        # replicas: same base, different local updates
        g = torch.Generator().manual_seed(0)
        base = torch.randn(4096, generator=g)
        model = base.clone()
        delta = torch.zeros(4096)
        delta[rank * 100:(rank + 1) * 100 + 50] = float(rank + 1)       # overlapping region 100..150
        model += delta
        sync = replicas.TorchReplicaSync(dist, mode=0)
        sync.sync(model, base_snapshot := base.clone())
        expect = base.clone()
        for r in range(world):
            d = torch.zeros(4096)
            d[r * 100:(r + 1) * 100 + 50] = float(r + 1)
            expect += d
        ok_delta = torch.allclose(model, expect, atol=1e-6) and torch.equal(base_snapshot, model)
        # average mode
        m2 = base.clone() + delta
        b2 = base.clone()
        replicas.TorchReplicaSync(dist, mode=1).sync(m2, b2)
        expect2 = base + sum((torch.zeros(4096).index_fill_(0, torch.arange(r * 100, (r + 1) * 100 + 50), float(r + 1))
                              for r in range(world))) / world
        ok_avg = torch.allclose(m2, expect2, atol=1e-6)
        q.put((rank, gathered, ok_delta, ok_avg))
    finally:
        dist.destroy_process_group()


def test_two_rank_replica_exchange_over_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, gathered, ok_delta, ok_avg in res:
        assert ok_delta and ok_avg
        offs = [g[0] for g in gathered]
        assert offs == [0, 4] and all(g[1] == 4 for g in gathered)
        allstarts = sum((g[2] for g in gathered), [])
        assert allstarts == [i * 125 for i in range(8)]


def test_world_size_one_is_a_bit_identical_noop():
    class FakeDist:
        @staticmethod
        def is_initialized():
            return False
    m = torch.randn(1000)
    keep = m.clone()
    b = torch.zeros(1000)
    replicas.TorchReplicaSync(FakeDist, 0).sync(m, b)
    assert torch.equal(m, keep)
    assert replicas.worker_plan(12, 1, 0) == (0, 12)
    with pytest.raises(ValueError):
        replicas.worker_plan(12, 8, 0)


def test_global_alpha_schedule_matches_reference_formula():
    # ref :391-392 in float32: alpha = a0 * (1 - wca / (iter*train_words + 1)), floored at a0*1e-4
    a = replicas.global_progress_alpha(0.05, 500_000, 5, 1_000_000)
    assert a == pytest.approx(0.05 * (1 - 500000 / 5000001), rel=1e-6)
    assert replicas.global_progress_alpha(0.05, 6_000_000, 5, 1_000_000) == pytest.approx(0.05 * 1e-4, rel=1e-6)
