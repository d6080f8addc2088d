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
        # mode 2 (round 6): saturation factor per row + quantization cells per element, against a single-process restatement
        V, D = 64, 8
        gg = torch.Generator().manual_seed(1)
        b3 = (torch.rand(2 * V * D, generator=gg) - 0.5)
        rate = torch.cat([torch.linspace(0.5, 1e-4, V), torch.linspace(0.2, 1e-4, V)])
        ds = []
        for r in range(world):
            gr = torch.Generator().manual_seed(10 + r)
            dr = (torch.rand(2 * V * D, generator=gr) - 0.5) * 2.0
            rows = torch.rand(2 * V, generator=gr) < (0.7 if r == 0 else 0.5)        # every replica touches its own subset of rows
            ds.append(dr * rows.to(dr.dtype).repeat_interleave(D))
        m3 = b3 + ds[rank]
        base3 = b3.clone()
        replicas.TorchReplicaSync(dist, mode=2, rate=rate, dim=D, bitlevel=1).sync(m3, base3, words=400)
        S = sum(ds)
        cnt = sum(((x.view(-1, D) != 0).any(1)).float() for x in ds)
        k = replicas.saturation_factors(rate, 400, cnt)
        safe = S * k.repeat_interleave(D)
        want = b3 + torch.where(((b3 + safe) < 0) == ((b3 + S) < 0), S, safe)
        both = cnt == 2
        ok_cells = (torch.allclose(m3, want, atol=1e-6) and torch.equal(m3, base3) and bool(both.any()) and bool((~both).any())
                    and float(k[both].min()) < 0.6 and bool((k[~both] == 1).all())
                    and 0.01 < float((((b3 + safe) < 0) != ((b3 + S) < 0)).float().mean()) < 0.5)
        q.put((rank, gathered, ok_delta, ok_avg and ok_cells))
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


# ---------------------------------------------------------------------------------------------------------------
# PhasedReplicaSync -- what bench.py --gpus N falls back to when the library's own communicator is not available, and
# what the one-GPU tests drive -- over gloo with a stand-in for the trainer that restates the library's exchange kernels
# (k_xchg_delta / k_xchg_touched / k_xchg_apply, w2b_kernels_misc.hip) on CPU tensors: the ORCHESTRATION is what is
# tested here (order of the phases, the chunks, the row counts of mode 2, the global word count).
class _FakeTrainer:
    def __init__(self, model, base, rows, dim, chunk, words):
        self.w, self.base = model, base.clone()            # base: the state all replicas agreed on at the last exchange
        self.rows, self.dim, self.chunk, self.words = rows, dim, chunk, words
        self.bufs, self.log, self.cnt, self.total = {}, [], None, None

    def _span(self, c):
        o = c * self.chunk
        return o, min(self.chunk, self.w.numel() - o)

    def exchange_begin(self):
        self.log.append("begin")
        self.cnt = None
        return (self.w.numel() + self.chunk - 1) // self.chunk, self.words

    def exchange_counts(self):
        self.log.append("counts")
        changed = (self.w.view(self.rows, self.dim) != self.base.view(self.rows, self.dim)).any(1)
        self.cnt = changed.to(torch.float32)
        self.bufs["cnt"] = self.cnt
        return "cnt", self.rows

    def exchange_delta(self, c):
        self.log.append("delta%d" % c)
        o, m = self._span(c)
        self.d = (self.w[o:o + m] - self.base[o:o + m]).clone()
        self.bufs[c] = self.d.clone()
        return c, m

    def exchange_apply(self, c, scale):
        self.log.append("apply%d" % c)
        o, m = self._span(c)
        s = self.bufs[c] * scale
        if self.cnt is not None:                      # contributor average (every row counts as saturated here)
            per_elem = self.cnt.clamp(min=1).repeat_interleave(self.dim)[o:o + m]
            s = s / per_elem
        self.w[o:o + m] += s - self.d
        self.base[o:o + m] += s

    def exchange_end(self, total):
        self.log.append("end")
        self.total = total

    def device_tensor(self, ptr, n):
        assert self.bufs[ptr].numel() == n
        return self.bufs[ptr]


def _phased_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows, dim, chunk = 40, 8, 96                  # 320 floats in chunks of 96: a ragged last chunk
        g = torch.Generator().manual_seed(0)
        base = torch.randn(rows * dim, generator=g)
        out = {}
        for mode in (0, 1, 2):
            model = base.clone()
            mine = torch.zeros(rows, dim)
            mine[rank * 10:rank * 10 + 15] = float(rank + 1)        # rows 10..14 are changed by both replicas
            model += mine.view(-1)
            t = _FakeTrainer(model, base, rows, dim, chunk, words=1000 * (rank + 1))
            replicas.PhasedReplicaSync(dist, t, mode).sync()
            out[mode] = (t.w.numpy().copy(), t.base.numpy().copy(), t.total, list(t.log))
        # numpy arrays travel by value; torch tensors would travel as file descriptors served by THIS process, which may
        # have exited by the time the parent unpickles them (FileNotFoundError on the resource-sharer socket: a flaky test)
        q.put((rank, base.numpy().copy(), out))
    finally:
        dist.destroy_process_group()


def test_phased_replica_sync_over_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_phased_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda x: x[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows, dim = 40, 8
    res = [(r, torch.from_numpy(b), {m: (torch.from_numpy(w), torch.from_numpy(bb), tot, log) for m, (w, bb, tot, log) in o.items()})
           for r, b, o in res]
    base = res[0][1]
    deltas = []
    for r in range(world):
        d = torch.zeros(rows, dim)
        d[r * 10:r * 10 + 15] = float(r + 1)
        deltas.append(d.view(-1))
    total = sum(deltas)
    cnt = sum((d.view(rows, dim) != 0).any(1).float() for d in deltas).clamp(min=1).repeat_interleave(dim)
    want = {0: base + total, 1: base + total / world, 2: base + total / cnt}
    for rank, _, out in res:
        for mode in (0, 1, 2):
            w, b, words, log = out[mode]
            assert torch.allclose(w, want[mode], atol=1e-6), (rank, mode)
            # base is the state the replicas agree on; a replica's own rows are (base + d) + (S - d): equal up to fp32 rounding
            assert torch.allclose(w, b, atol=1e-6) and torch.allclose(b, want[mode], atol=1e-6)
            assert words == 1000 + 2000                                # the alpha schedule's global word count (ref :391)
            phases = ["begin"] + (["counts"] if mode == 2 else []) + [x for c in range(4) for x in ("delta%d" % c, "apply%d" % c)] + ["end"]
            assert log == phases
    assert torch.equal(res[0][2][2][1], res[1][2][2][1])               # `base` is bit-identical across the ranks
