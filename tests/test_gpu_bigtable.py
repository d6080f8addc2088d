"""-m gpu: the LARGE-TABLE form of the row accesses (tables >= 2 GiB: per-row buffer descriptors, `tab_bytes == 0`,
w2b_device.hpp load_col/store_col; the `MM + 8` instantiations of the sentence-resident kernel).

Three layers:
  1. the bit-exact parity tests of the small-table form re-run with w2b_tuning.force_row_desc = 1 (selects the
     large-table form on any table size; `./word2bits -row-desc 1`) -- tuple form, plain worker kernel,
     sentence-resident kernel;
  2. a genuine > 2 GiB table (V = 700 000 x D = 800: 2.24 GB per table) -- collision-free tuple batch against the
     oracle on the touched rows (bit-exact in parity mode, rounding-tight in the fast mode), every other row still
     holding its InitNet bits, and sentence-resident == plain worker kernel bit for bit on the whole 4.5 GB model;
  3. BASELINE configs[4] shape (V = 3.7 M, D = 1000, negative 12, bitlevel 0 and 1: 14.8 GB per table), same checks.
The oracle cannot hold a 3.7 M x 1000 model twice in host memory next to the test, so the touched rows are mapped to
a compact vocabulary whose rows are initialised with the InitNet values of the real rows (ref src/word2bits.cpp:343-361:
the k-th draw of the LCG seeded with 1; v is filled first, then u)."""
import numpy as np
import pytest

import word2bits_amd as w2b
from w2b_testlib import OracleState

import test_gpu_exact
import test_gpu_worker

pytestmark = pytest.mark.gpu


@pytest.fixture
def row_desc(monkeypatch):
    monkeypatch.setitem(w2b.Trainer.default_tuning, "force_row_desc", 1)          # every Trainer the re-used tests create
    monkeypatch.setattr(test_gpu_worker, "EXTRA_CLI", ["-row-desc", "1"])          # ... and every ./word2bits they run


# ------------------------------------------------------------------------------- 1. forced form on small tables
@pytest.mark.parametrize("D,window,negative,bitlevel,reg", [
    (800, 8, 24, 1, 0.0), (400, 8, 24, 2, 0.0), (1000, 5, 12, 0, 0.0), (100, 5, 5, 4, 0.001), (50, 4, 3, 1, 0.0),
    (1200, 2, 3, 2, 0.0), (36, 40, 70, 1, 0.0), (1, 3, 2, 2, 0.0),
])
def test_tuple_updates_bit_exact_row_desc(gpu, row_desc, D, window, negative, bitlevel, reg):
    test_gpu_exact.test_tuple_updates_bit_exact(gpu, D, window, negative, bitlevel, reg)


@pytest.mark.parametrize("bitlevel,sample,D,window,negative,iters", [
    (1, 1e-3, 200, 8, 24, 1), (2, 0.0, 100, 3, 7, 1), (1, 0.0, 800, 8, 24, 1),
])
def test_single_worker_epochs_bit_exact_row_desc(gpu, row_desc, bitlevel, sample, D, window, negative, iters):
    test_gpu_exact.test_single_worker_epochs_bit_exact(gpu, bitlevel, sample, D, window, negative, iters)


@pytest.mark.parametrize("D,window,negative,bitlevel", [(800, 8, 24, 1), (200, 8, 24, 2), (64, 2, 3, 0), (768, 12, 5, 1)])
def test_resident_equals_plain_single_worker_row_desc(gpu, row_desc, D, window, negative, bitlevel, monkeypatch):
    test_gpu_worker.test_sentence_resident_kernel_equals_plain_kernel_single_worker(gpu, D, window, negative, bitlevel,
                                                                                    None, monkeypatch)


@pytest.mark.parametrize("threads,size,window,bitlevel", [(16, 200, 8, 1), (5, 800, 8, 0)])
def test_resident_equals_plain_many_workers_row_desc(gpu, row_desc, threads, size, window, bitlevel, tmp_path):
    test_gpu_worker.test_sentence_resident_kernel_equals_plain_kernel_many_workers(gpu, threads, size, window, bitlevel,
                                                                                   tmp_path)


# ------------------------------------------------------------------------------- helpers for genuine big tables
def init_lut():
    """value of the k-th InitNet draw (ref :350-360); the low 16 bits of the LCG have period 65536"""
    lut = np.empty(65536, np.float32)
    x = np.uint64(1)
    a, c = np.uint64(25214903917), np.uint64(11)
    with np.errstate(over="ignore"):
        for k in range(65536):
            x = x * a + c
            lut[k] = np.float32(np.float32(int(x) & 0xFFFF) / np.float32(65536)) - np.float32(0.5)
    return lut


def init_rows(lut, V, D, rows, table):
    """InitNet values of `rows` of table 'v' (filled first) or 'u'"""
    n = V * D
    base = (0 if table == "v" else n) + rows.astype(np.int64)[:, None] * D + np.arange(D, dtype=np.int64)[None, :]
    return lut[base & 65535]


def spread_rows(rng, V, D, count):
    """distinct rows in [1, V) that include the first and last rows and the rows on both sides of every 2 GiB line
    of a table (where 32-bit offsets would wrap)"""
    special = {1, 2, V - 1, V - 2}
    for gib2 in range(1, int(V * D * 4 // (1 << 31)) + 1):
        r = (gib2 << 31) // (D * 4)
        special |= {r - 1, r, r + 1}
    special = np.array(sorted(x for x in special if 1 <= x < V), np.int64)
    rest = rng.choice(np.arange(1, V, dtype=np.int64), size=count, replace=False)
    rest = rest[~np.isin(rest, special)]
    out = np.concatenate([special, rest])[:count]
    return rng.permutation(out)


def device_tables(t, V, D):
    import torch
    flat = t.model_tensor()
    return flat[:V * D].view(V, D), flat[V * D:].view(V, D)


def changed_rows(tab, lut_dev, V, D, first_index):
    """rows of a device table that no longer hold their InitNet bits (chunked, on the device)"""
    import torch
    out = []
    step = max(1, (1 << 27) // D)
    cols = torch.arange(D, device=tab.device, dtype=torch.int64)
    for r0 in range(0, V, step):
        r1 = min(V, r0 + step)
        idx = (first_index + torch.arange(r0, r1, device=tab.device, dtype=torch.int64)[:, None] * D + cols[None, :]) & 65535
        bad = (tab[r0:r1].view(torch.int32) != lut_dev[idx].view(torch.int32)).any(dim=1)
        out.append(torch.nonzero(bad).flatten() + r0)
    return torch.cat(out).cpu().numpy()


def big_table_tuple_check(V, D, window, negative, bitlevel, exact, n=48, seed=0):
    import torch
    rng = np.random.default_rng(seed)
    lut = init_lut()
    t = w2b.Trainer(V, D, window, negative, bitlevel, num_threads=1, iter=1, alpha=0.05, sample=0.0,
                    train_words=10 ** 9, compute_loss=False, exact=exact)
    t.init_net()
    cw_all = [int(rng.integers(1, 2 * window + 1)) for _ in range(n)]
    rows_u = spread_rows(rng, V, D, sum(cw_all))
    rows_v = spread_rows(rng, V, D, n * (negative + 1))
    center = rows_v[:n].astype(np.int32)
    neg = rows_v[n:].reshape(n, negative).astype(np.int32)
    ctx = rows_u.astype(np.int32)
    ctx_off = np.concatenate([[0], np.cumsum(cw_all)]).astype(np.int32)
    # a duplicate context word and a duplicate negative inside some tuples (ref :494-503, :450-491)
    for i in range(0, n, 3):
        if cw_all[i] >= 3:
            ctx[ctx_off[i + 1] - 1] = ctx[ctx_off[i]]
        if negative >= 3:
            neg[i, 2] = neg[i, 0]
    # compact oracle: row k+1 of the oracle <-> k-th distinct touched row
    uu, uinv = np.unique(ctx, return_inverse=True)
    vv, vinv = np.unique(np.concatenate([center, neg.ravel()]), return_inverse=True)
    Vc = max(len(uu), len(vv)) + 1
    o = OracleState(np.full(Vc, 10, np.int64), D, window=window, negative=negative, bitlevel=bitlevel, sample=0.0,
                    table_size=1000, init=False)
    o.u[1:len(uu) + 1] = init_rows(lut, V, D, uu, "u")
    o.v[1:len(vv) + 1] = init_rows(lut, V, D, vv, "v")
    c_center = (vinv[:n] + 1).astype(np.int32)
    c_neg = (vinv[n:] + 1).reshape(n, negative).astype(np.int32)
    c_ctx = (uinv + 1).astype(np.int32)
    for rep in range(2 if exact else 1):     # (fast mode: one step, as in tests/test_gpu_parity.py -- a flipped sigmoid
        o.train_tuples(c_center, ctx_off, c_ctx, c_neg, 0.05)     # bin would compound over a second one)
        t.train_tuples(center, ctx_off, ctx, neg, 0.05)
    du, dv = device_tables(t, V, D)
    gu = du[torch.from_numpy(uu).to(du.device)].cpu().numpy()
    gv = dv[torch.from_numpy(vv).to(dv.device)].cpu().numpy()
    wu, wv = o.u[1:len(uu) + 1], o.v[1:len(vv) + 1]
    if exact:
        assert np.array_equal(gu.view(np.uint32), wu.view(np.uint32)), np.abs(gu - wu).max()
        assert np.array_equal(gv.view(np.uint32), wv.view(np.uint32)), np.abs(gv - wv).max()
    else:
        # two updates of the same rows with the dot product re-associated: a neighbouring sigmoid bin moves g by one
        # table step (tests/test_gpu_parity.py); everything else agrees to rounding
        tol = 3 * 1.6e-4 * {0: 0.6, 1: 1.0 / 3, 2: 0.75}[bitlevel] + 2e-6
        assert np.abs(gu - wu).max() <= tol and np.abs(gv - wv).max() <= tol
        assert np.mean(np.abs(gv - wv).max(axis=1) <= 4e-6) > 0.9
    # every row the batch did not name still holds its InitNet bits; the named ones moved
    lut_dev = torch.from_numpy(lut).to(du.device)
    cu = changed_rows(du, lut_dev, V, D, V * D)
    cv = changed_rows(dv, lut_dev, V, D, 0)
    assert set(cu.tolist()) <= set(uu.tolist()) and len(cu) >= 0.99 * len(uu)
    assert set(cv.tolist()) <= set(vv.tolist()) and len(cv) >= 0.99 * len(vv)
    t.close()


def big_table_worker_check(V, D, window, negative, bitlevel, positions=2500, seed=1):
    """one Hogwild worker over a token stream whose ids cover the whole table: the sentence-resident kernel and the
    plain kernel must leave bit-identical models (compared on the device), both in the large-table form"""
    import os
    import torch
    rng = np.random.default_rng(seed)
    ids = spread_rows(rng, V, D, positions).astype(np.int32)
    ids[40::41] = 0
    cn = np.ones(V, np.int64)
    cn[ids[ids > 0]] += 50
    try:
        models = []
        for wc in (True, False):
            t = w2b.Trainer(V, D, window, negative, bitlevel, num_threads=1, iter=1, sample=0.0,
                            train_words=int(len(ids)), compute_loss=False, window_cache=wc)
            t.init_net()
            t.set_vocab_counts(cn, 200000)
            t.set_corpus(ids)
            t.set_shards(np.zeros(1, np.int64))
            t.train_epoch(positions_per_launch=700)
            models.append(t)
        a, b = models[0].model_tensor(), models[1].model_tensor()
        same = True
        for o in range(0, a.numel(), 1 << 28):
            same = same and bool(torch.equal(a[o:o + (1 << 28)].view(torch.int32), b[o:o + (1 << 28)].view(torch.int32)))
        lut_dev = torch.from_numpy(init_lut()).to(a.device)
        moved = len(changed_rows(device_tables(models[0], V, D)[0], lut_dev, V, D, V * D))
        for t in models:
            t.close()
        assert same
        assert moved >= 0.9 * len(np.unique(ids[ids > 0]))      # the run really trained rows all over the table
    finally:
        pass


# ------------------------------------------------------------------------------- 2. a genuine > 2 GiB table
@pytest.mark.parametrize("exact", [True, False])
def test_table_over_2gib_tuples(gpu, exact):
    big_table_tuple_check(700_000, 800, 8, 24, 1, exact)


def test_table_over_2gib_resident_equals_plain(gpu):
    big_table_worker_check(700_000, 800, 8, 24, 1)


# ------------------------------------------------------------------------------- 3. BASELINE configs[4] shape
@pytest.mark.parametrize("bitlevel,exact", [(0, True), (1, True), (1, False)])
def test_cfg5_shape_tuples(gpu, bitlevel, exact):
    """V = 3.7 M, D = 1000, negative 12 (BASELINE configs[4]): 14.8 GB per table"""
    big_table_tuple_check(3_700_000, 1000, 8, 12, bitlevel, exact, n=32)


@pytest.mark.parametrize("bitlevel", [0, 1])
def test_cfg5_shape_resident_equals_plain(gpu, bitlevel):
    probe = w2b.Trainer(2, 1000, 8, 12, bitlevel, num_threads=1, compute_loss=False, window_cache=True)
    resident = probe.worker_kernel_info()[0]
    probe.close()
    assert resident, "no sentence-resident kernel for the configs[4] shape (D=1000, window 8, negative 12)"
    big_table_worker_check(3_700_000, 1000, 8, 12, bitlevel, positions=1500)
