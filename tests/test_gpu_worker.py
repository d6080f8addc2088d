"""-m gpu: form (i) -- the on-device restatement of TrainModelThread (sentence reader, sub-sampling,
window draw, negative draws, alpha schedule) against the CPU oracle on the same token stream."""
import numpy as np
import pytest

import word2bits_amd as w2b
from w2b_testlib import OracleState, zipf_ids

pytestmark = pytest.mark.gpu
EXTRA_CLI = []          # extra ./word2bits arguments (test_gpu_bigtable.py re-runs tests of this file with -row-desc 1)


def token_stream(rng, V, n, line=37):
    ids = zipf_ids(rng, V, n).astype(np.int32)
    ids[line::line] = 0                      # "</s>" every `line` tokens
    ids[5] = 0
    ids[6] = 0                               # an empty sentence (ref :428 still draws)
    return ids


def counts_of(ids, V):
    cn = np.bincount(ids, minlength=V).astype(np.int64)
    cn[cn == 0] = 1
    return cn


def drift(a, b):
    return float(np.abs(a - b).mean()), float(np.mean(np.signbit(a) != np.signbit(b)))


def setup(V, ids, D, window, negative, bitlevel, sample, iters, num_threads=1, fma=False):
    cn = counts_of(ids, V)
    tw = int(cn.sum())
    o = OracleState(cn, D, window=window, negative=negative, bitlevel=bitlevel, num_threads=num_threads,
                    iters=iters, sample=sample, table_size=50000, fma=fma)
    o.m.train_words = tw
    return cn, tw, o


@pytest.mark.parametrize("bitlevel,sample,D,window,negative", [
    (1, 0.0, 64, 5, 5),
    (1, 1e-3, 200, 8, 24),
    (0, 1e-3, 200, 8, 24),
    (2, 0.0, 100, 3, 7),
    (1, 1e-3, 32, 1, 0),       # -window 1 -negative 0
])
@pytest.mark.parametrize("window_cache", [True, False])
def test_single_worker_short_horizon_tight(gpu, bitlevel, sample, D, window, negative, window_cache):
    """3000 positions over a 5000-word vocabulary: rows are rarely revisited, so the worker form
    (on-device sentence reader / window / negative draws / alpha) must track the oracle closely."""
    V, n = 5000, 3000
    rng = np.random.default_rng(4)
    ids = token_stream(rng, V, n)
    cn, tw, o = setup(V, ids, D, window, negative, bitlevel, sample, 1)
    _, _, y = setup(V, ids, D, window, negative, bitlevel, sample, 1, fma=True)
    t = w2b.Trainer(V, D, window, negative, bitlevel, num_threads=1, iter=1, sample=sample, train_words=tw,
                    window_cache=window_cache)
    t.set_model(o.u, o.v)
    t.set_vocab_counts(cn, 50000)
    t.set_corpus(ids)
    t.set_shards(np.zeros(1, np.int64))
    lo = o.train_epoch_tokens(ids, np.zeros(1, np.int64))
    y.train_epoch_tokens(ids, np.zeros(1, np.int64))
    lg = t.train_epoch(positions_per_launch=501)
    fin, wca, alpha, _ = t.epoch_status()
    assert fin and wca == o.m.word_count_actual and np.float32(alpha) == np.float32(o.m.alpha)
    u, v = t.get_model()
    for got, ref, yard in ((u, o.u, y.u), (v, o.v, y.v)):
        gm, gs = drift(got, ref)
        ym, ys = drift(yard, ref)
        # no farther than 3x the reference's own FMA build; the floor covers the case where the yardstick
        # run happened to see no level flip at all (a single flip of a quantized level moves ~1e-3 mean)
        floor = 1e-5 if bitlevel == 0 else 2e-3
        assert gm <= 3 * ym + floor, (gm, ym)
        assert gs <= 3 * ys + 2e-3, (gs, ys)
    assert lg == pytest.approx(lo, rel=2e-3)
    t.close()


@pytest.mark.parametrize("bitlevel,sample,D,window,negative", [
    (1, 1e-3, 64, 5, 5),
    (0, 1e-2, 48, 3, 7),
    (2, 0.0, 100, 8, 24),
    (1, 0.0, 800, 8, 24),
])
@pytest.mark.parametrize("window_cache", [True, False])
def test_single_worker_long_horizon_statistical(gpu, bitlevel, sample, D, window, negative, window_cache):
    """2 epochs x 30000 tokens on 150 rows: every row is rewritten thousands of times and quantized
    training is chaotic (two builds of the REFERENCE disagree on 4-25% of the signs here), so only
    the integer bookkeeping is exact; values are held to 'no farther from the bit-reference than
    1.5x the drift of the oracle's own FMA build' and the epoch loss to 2%."""
    V, n = 150, 30000
    rng = np.random.default_rng(11)
    ids = token_stream(rng, V, n)
    ids[2000:3300] = zipf_ids(rng, V, 1300)  # one 1300-token line: sentence chunking at 1000 (ref :410)
    cn, tw, o = setup(V, ids, D, window, negative, bitlevel, sample, 2)
    _, _, y = setup(V, ids, D, window, negative, bitlevel, sample, 2, fma=True)
    t = w2b.Trainer(V, D, window, negative, bitlevel, num_threads=1, iter=2, alpha=0.05, sample=sample,
                    train_words=tw, compute_loss=True, window_cache=window_cache)
    t.set_model(o.u, o.v)
    t.set_vocab_counts(cn, 50000)
    t.set_corpus(ids)
    t.set_shards(np.zeros(1, np.int64))
    for ep in range(2):
        lo = o.train_epoch_tokens(ids, np.zeros(1, np.int64))
        y.train_epoch_tokens(ids, np.zeros(1, np.int64))
        lg = t.train_epoch(positions_per_launch=777)     # odd launch size: exercises save/restore
        fin, wca, alpha, _ = t.epoch_status()
        assert fin
        assert wca == o.m.word_count_actual                      # integer bookkeeping: exact
        assert np.float32(alpha) == np.float32(o.m.alpha)        # alpha staircase: exact
        assert lg == pytest.approx(lo, rel=2e-2)
    u, v = t.get_model()
    assert np.isfinite(u).all() and np.isfinite(v).all()
    for got, ref, yard in ((u, o.u, y.u), (v, o.v, y.v)):
        gm, gs = drift(got, ref)
        ym, ys = drift(yard, ref)
        assert gm <= 1.5 * ym + 1e-4, (gm, ym)
        assert gs <= 1.5 * ys + 1e-3, (gs, ys)
    t.close()


@pytest.mark.parametrize("window_cache", [True, False])
def test_shard_override_and_multi_worker_bookkeeping(gpu, window_cache):
    """4 workers (Hogwild): values are racy, but the integer bookkeeping is deterministic."""
    V, n, D = 120, 20000, 32
    rng = np.random.default_rng(2)
    ids = token_stream(rng, V, n)
    cn = counts_of(ids, V)
    tw = int(cn.sum())
    starts = np.array([0, 5000, 10001, 15002], np.int64)
    ov = np.array([-2, 7, -1, 0], np.int32)      # truncated first word: in vocab / OOV / "</s>"
    o = OracleState(cn, D, window=5, negative=5, bitlevel=1, num_threads=4, iters=1, sample=1e-3,
                    table_size=50000)
    o.m.train_words = tw
    t = w2b.Trainer(V, D, 5, 5, 1, num_threads=4, iter=1, sample=1e-3, train_words=tw, window_cache=window_cache)
    t.set_model(o.u, o.v)
    t.set_vocab_counts(cn, 50000)
    t.set_corpus(ids)
    t.set_shards(starts, ov)
    o_loss = o.train_epoch_tokens(ids, starts, ov)
    t.train_epoch(positions_per_launch=500)
    fin, wca, alpha, loss = t.epoch_status()
    assert fin and wca == o.m.word_count_actual
    u, v = t.get_model()
    assert np.isfinite(u).all() and np.isfinite(v).all()
    assert loss == pytest.approx(o_loss, rel=5e-2)            # racy values, same objective
    t.close()


@pytest.mark.parametrize("D,window,negative,bitlevel", [
    (800, 8, 24, 1), (200, 8, 24, 2), (64, 2, 3, 0), (96, 1, 2, 1),
    (1000, 8, 12, 1),        # BASELINE configs[4] row length: the whole window still fits next to a second workgroup
    (768, 12, 5, 1),         # window too wide for LDS: radius window-1, the outermost context rows are register-held
    (1024, 3, 3, 2),         # the widest row of the 16-byte-column form
])
@pytest.mark.parametrize("hot", [0, None, 8])
def test_sentence_resident_kernel_equals_plain_kernel_single_worker(gpu, D, window, negative, bitlevel, hot, monkeypatch):
    """The LDS-resident window is an optimisation, not a different algorithm: with one worker nobody else
    touches a resident row, so the exact fp32 value is written back and the whole model must come out
    BIT-IDENTICAL to the plain kernel (same launches, same dot-product reduction tree, same target order) --
    without per-XCD copies of hot rows (w2b_tuning.hot_rows_* = 0), with the number the library derives from the word
    counts and the worker count (default: none for one worker) and with 8 forced, merged every 2 steps."""
    tune = {} if hot is None else dict(hot_rows_v=hot, hot_rows_u=hot, hot_period=2)
    V, n = 300, 6000
    rng = np.random.default_rng(9)
    ids = token_stream(rng, V, n, line=23)          # short sentences: many window fills/flushes
    ids[3000:4100] = zipf_ids(rng, V, 1100)          # and one sentence longer than 1000 tokens
    cn = counts_of(ids, V)
    tw = int(cn.sum())
    out = []
    for wc, pos in ((True, 333), (False, 333), (True, 50)):
        t = w2b.Trainer(V, D, window, negative, bitlevel, num_threads=1, iter=1, sample=1e-3, train_words=tw,
                        window_cache=wc, **tune)
        t.init_net()
        t.set_vocab_counts(cn, 50000)
        t.set_corpus(ids)
        t.set_shards(np.zeros(1, np.int64))
        if wc:
            resident, radius, colb, _, nh = t.worker_kernel_info()
            assert resident and colb == 16 and radius == (window - 1 if (D, window) == (768, 12) else window)
            assert nh == (8 if hot == 8 else 0)
        loss = t.train_epoch(positions_per_launch=pos)
        u, v = t.get_model()
        out.append((u, v, loss, t.epoch_status()[1]))
        t.close()
    for k in (1, 2):
        assert out[0][3] == out[k][3]
        assert np.array_equal(out[0][0].view(np.uint32), out[k][0].view(np.uint32))
        assert np.array_equal(out[0][1].view(np.uint32), out[k][1].view(np.uint32))
        assert out[0][2] == pytest.approx(out[k][2], rel=1e-9)


@pytest.mark.parametrize("threads,size,window,bitlevel", [(16, 200, 8, 1), (8, 64, 3, 2), (5, 800, 8, 0)])
def test_sentence_resident_kernel_equals_plain_kernel_many_workers(gpu, threads, size, window, bitlevel, tmp_path):
    """Several Hogwild workers, deterministic anyway (shards with disjoint vocabularies, -negative 0, shards shorter
    than an alpha period: no two workers share a row): the sentence-resident kernel -- window slots, scratch entries,
    exact-or-merge write-back, per-XCD copies of the six hottest rows merged every 4 steps (every worker is the
    only writer of its rows, whichever XCD it runs on), producer wavefront -- must write the same file as the plain
    kernel, byte for byte."""
    import os
    import subprocess
    from w2b_testlib import ROOT, write_disjoint_shard_corpus
    corpus = write_disjoint_shard_corpus(str(tmp_path / "c.txt"), n_shards=threads, sentences=80, seed=threads)
    outs = []
    for wc in (0, 1):
        out = str(tmp_path / ("o%d.vec" % wc))
        r = subprocess.run([os.path.join(ROOT, "word2bits"), "-train", corpus, "-output", out, "-threads", str(threads),
                            "-window-cache", str(wc), "-bitlevel", str(bitlevel), "-size", str(size), "-window", str(window),
                            "-negative", "0", "-iter", "2", "-min-count", "1", "-binary", "1", "-sample", "0",
                            "-positions", "53", "-hot-rows", "6", "-hot-period", "4"] + EXTRA_CLI,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-300:]
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1]


@pytest.mark.parametrize("window_cache", [True, False])
def test_epoch_poll_one_launch_behind(gpu, window_cache):
    """w2b_epoch_poll(lag=1): the host looks at the launch before the latest one without waiting for the latest -- the
    loop ./word2bits runs.  The epoch is seen finished one launch late, the counters are the blocking call's, the
    device-accumulated loss equals the per-worker sum up to the order of the additions."""
    V, n, D = 200, 30000, 64
    rng = np.random.default_rng(3)
    ids = token_stream(rng, V, n)
    cn = counts_of(ids, V)
    t = w2b.Trainer(V, D, 5, 5, 1, num_threads=6, iter=1, sample=1e-3, train_words=int(cn.sum()), window_cache=window_cache)
    t.init_net()
    t.set_vocab_counts(cn, 50000)
    t.set_corpus(ids)
    t.set_shards(np.arange(6, dtype=np.int64) * (n // 6))
    for epoch in range(2):
        t.epoch_begin()
        assert t.epoch_poll(1) == (False, 0, pytest.approx(0.05), 0.0)        # nothing launched yet
        launches, seen_wca = 0, -1
        while True:
            t.train_step(400)
            launches += 1
            fin, wca, alpha, loss = t.epoch_poll(1)
            assert wca >= seen_wca
            seen_wca = wca
            if fin:
                break
            assert launches < 200
        fin0, wca0, alpha0, loss0 = t.epoch_poll(0)
        fin2, wca2, alpha2, loss2 = t.epoch_status()
        assert fin0 and fin2 and wca0 == wca2 == wca and alpha0 == alpha2
        assert loss0 == pytest.approx(loss2, rel=1e-9) and loss == pytest.approx(loss2, rel=1e-9) and loss2 < 0
    t.close()


# ------------------------------------------------------------------ round 4: lossless rows / fresh rows / late round
@pytest.mark.parametrize("knobs", [
    dict(atomic_rank=149, atomic_rank_u=149),                 # every row's update an atomic add (transposed 16-byte-column form)
    dict(atomic_rank=40, atomic_rank_u=-1),                   # v only, a prefix
    dict(fresh_rank_u=149),                                   # context rows re-read before their update
    dict(hot_rows_v=8, hot_rows_u=0, atomic_rank_u=149, hot_period=2),     # copies of hot target rows + lossless context rows
    dict(hot_rows_v=8, hot_rows_u=4, fresh_rank_u=60, atomic_rank=60, atomic_rank_u=60, hot_period=4),
])
@pytest.mark.parametrize("D,bitlevel,loss", [(800, 1, True), (200, 2, False), (36, 0, True)])
def test_single_worker_is_bit_identical_under_every_round4_knob(gpu, knobs, D, bitlevel, loss):
    """One worker has nobody to race with: an atomic add of d lands as fl(x + d), a re-read row is the row it read -- so the plain kernel with any of the round-4 knobs must leave
    the very bits it leaves without them (u, v, word count, alpha, epoch loss).  D = 36: partly filled wavefront."""
    V, n = 150, 12000
    rng = np.random.default_rng(21)
    ids = token_stream(rng, V, n)
    cn = counts_of(ids, V)
    res = []
    for kw in ({}, knobs):
        t = w2b.Trainer(V, D, 8, 24, bitlevel, num_threads=1, iter=1, sample=0.0, train_words=int(cn.sum()),
                        compute_loss=loss, window_cache=False, **kw)
        t.init_net()
        t.set_vocab_counts(cn, 50000)
        t.set_corpus(ids)
        t.set_shards(np.zeros(1, np.int64))
        lg = t.train_epoch(positions_per_launch=611)
        fin, wca, alpha, _ = t.epoch_status()
        u, v = t.get_model()
        res.append((u, v, wca, alpha, lg))
        t.close()
    (u0, v0, w0, a0, l0), (u1, v1, w1, a1, l1) = res
    assert np.array_equal(u0.view(np.uint32), u1.view(np.uint32)) and np.array_equal(v0.view(np.uint32), v1.view(np.uint32))
    assert w0 == w1 and a0 == a1 and l0 == l1
