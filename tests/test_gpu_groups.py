"""-m gpu: the ROW-GROUP worker kernel (word2bits_amd/csrc/w2b_kernels_groups.hip, round 5) against the plain worker kernel.

The row-group kernel spreads the rows of a centre word over G groups of wavefronts, prepares the lists one word ahead on a
producer wavefront and lets an adder wavefront issue the lossless adds to the frequent context rows.  It is a different
SCHEDULE of the same arithmetic: the window average is summed in window order, the error in target order, the dot product
uses the plain kernel's tree, repeated target rows are taken again after their first update.  So with one worker -- and
with several workers that never share a row -- it must leave the plain kernel's bits: u, v, word count, alpha, epoch loss.
The plain kernel in turn is pinned to the oracle / the unmodified reference (tests/test_gpu_exact.py, test_gpu_worker.py)."""
import os
import subprocess

import numpy as np
import pytest

import word2bits_amd as w2b
from w2b_testlib import ROOT, zipf_ids, write_disjoint_shard_corpus
from test_gpu_worker import token_stream, counts_of

pytestmark = pytest.mark.gpu


def run(V, ids, cn, D, window, negative, bitlevel, pos, groups, sample=1e-3, reg=0.0, loss=True, threads=1, starts=None, **tune):
    t = w2b.Trainer(V, D, window, negative, bitlevel, num_threads=threads, iter=1, sample=sample, reg=reg,
                    train_words=int(cn.sum()), compute_loss=loss, row_groups=groups, **tune)
    t.init_net()
    t.set_vocab_counts(cn, 50000)
    t.set_corpus(ids)
    t.set_shards(np.zeros(1, np.int64) if starts is None else starts)
    name = t.worker_kernel_name()
    lg = t.train_epoch(positions_per_launch=pos)
    fin, wca, alpha, _ = t.epoch_status()
    u, v = t.get_model()
    t.close()
    return name, u, v, wca, alpha, lg


@pytest.mark.parametrize("D,window,negative,bitlevel", [
    (200, 8, 24, 1),          # BASELINE configs[0] row length: one wavefront per row, 4 groups
    (400, 8, 24, 2),          # configs[2] row length: two wavefronts per row
    (800, 8, 24, 1),          # configs[1] row length: four wavefronts per row, 3 groups
    (36, 5, 5, 0),            # partly filled wavefront
    (64, 2, 3, 4),            # generic quantizer
    (256, 16, 27, 1),         # widest window, most negatives of the one-wavefront form
    (512, 8, 12, 0),
    (1024, 3, 26, 2),         # widest row, most negatives of the four-wavefront form
    (300, 1, 1, 1),
])
@pytest.mark.parametrize("knobs", [dict(), dict(atomic_rank=299, atomic_rank_u=299), dict(atomic_rank_u=25, atomic_rank=0)])
def test_row_group_kernel_equals_plain_kernel_single_worker(gpu, D, window, negative, bitlevel, knobs):
    """300 words, Zipf ids: hot rows are context rows of consecutive positions (handled by different groups from one word to
    the next), targets repeat inside a centre word, the centre word is drawn as its own negative; short sentences, one of
    1100 tokens, an empty one; odd launch lengths (save / restore of the worker, the producer's one-word lead).  Every row
    lossless (the adder wavefront for u, transposed adds for v), a prefix of u only, or none."""
    V, n = 300, 6000
    rng = np.random.default_rng(9)
    ids = token_stream(rng, V, n, line=23)
    ids[3000:4100] = zipf_ids(rng, V, 1100)
    cn = counts_of(ids, V)
    ref = run(V, ids, cn, D, window, negative, bitlevel, 333, False, **knobs)
    assert ref[0] == "plain"
    for pos in (333, 50):
        got = run(V, ids, cn, D, window, negative, bitlevel, pos, True, **knobs)
        assert got[0] == "groups"
        assert got[3] == ref[3] and got[4] == ref[4]
        assert np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32))
        assert np.array_equal(got[2].view(np.uint32), ref[2].view(np.uint32))
        assert got[5] == ref[5]                      # the log-sigmoid terms are booked lane by lane as in the plain kernel


@pytest.mark.parametrize("D,bitlevel,knobs", [(200, 1, dict()), (400, 2, dict(atomic_rank=149, atomic_rank_u=149)), (800, 0, dict())])
def test_row_group_kernel_with_regularisation(gpu, D, bitlevel, knobs):
    """-reg != 0: the delta of a context row depends on the row (the data wavefronts do those adds themselves) and the
    reg * sum q^2 terms of the loss are summed per group -- values bit-identical, loss up to the order of the additions"""
    V, n = 150, 5000
    rng = np.random.default_rng(5)
    ids = token_stream(rng, V, n)
    cn = counts_of(ids, V)
    ref = run(V, ids, cn, D, 8, 24, bitlevel, 400, False, sample=0.0, reg=1e-3, **knobs)
    got = run(V, ids, cn, D, 8, 24, bitlevel, 400, True, sample=0.0, reg=1e-3, **knobs)
    assert (ref[0], got[0]) == ("plain", "groups")
    assert np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32))
    assert np.array_equal(got[2].view(np.uint32), ref[2].view(np.uint32))
    assert got[5] == pytest.approx(ref[5], rel=1e-6)


def test_row_group_kernel_falls_back_where_it_does_not_fit(gpu):
    """more negatives than the groups hold, a window wider than 16, 4-byte columns, relaxed rows, the parity mode: the plain
    kernel runs, and w2b_worker_kernel_info says so"""
    V = 200
    for kw in (dict(negative=30), dict(window=17), dict(layer1_size=202), dict(relaxed_coherence=True), dict(exact=True),
               dict(hot_rows_v=4), dict(fresh_rank_u=10)):
        args = dict(layer1_size=200, window=8, negative=24)
        args.update(kw)
        t = w2b.Trainer(V, args.pop("layer1_size"), args.pop("window"), args.pop("negative"), 1, num_threads=4, train_words=10000,
                        row_groups=True, **args)
        t.set_vocab_counts(np.full(V, 50, np.int64), 0)
        assert t.worker_kernel_name() == "plain", kw
        t.close()
    # automatic: short rows run the row groups -- unless the fidelity budget is thin already (w2b_trainer.cpp groups_plan):
    # shards shorter than 50 000 words per worker, or a vocabulary so small and flat that every row collides
    Vz = 60000
    zipf = np.maximum(5, (3e7 / (np.arange(Vz) + 1.0))).astype(np.int64)
    zipf[0] = 0
    for kw, want in ((dict(), "groups"), (dict(layer1_size=800), "plain"), (dict(train_words=256 * 40000), "plain"),
                     (dict(flat=True, train_words=10 ** 9), "plain")):
        D = kw.get("layer1_size", 200)
        cn = np.full(2000, 300, np.int64) if kw.get("flat") else zipf
        t = w2b.Trainer(len(cn), D, 8, 24, 1, num_threads=256, train_words=kw.get("train_words", int(cn.sum())))
        t.set_vocab_counts(cn, 0)
        assert t.worker_kernel_name() == want, kw
        t.close()
    t = w2b.Trainer(V, 200, 8, 24, 1, num_threads=4, train_words=10000, row_groups=False)
    assert t.worker_kernel_name() == "plain"
    t.close()


@pytest.mark.parametrize("threads,size,window,bitlevel", [(16, 200, 8, 1), (8, 400, 3, 2), (5, 800, 8, 0)])
def test_row_group_kernel_equals_plain_kernel_many_workers(gpu, threads, size, window, bitlevel, tmp_path):
    """Several Hogwild workers, deterministic anyway (shards with disjoint vocabularies, -negative 0, shards shorter than
    an alpha period): the same file as the plain kernel, byte for byte, through the command line"""
    corpus = write_disjoint_shard_corpus(str(tmp_path / "c.txt"), n_shards=threads, sentences=80, seed=threads)
    outs = []
    for flag in (["-row-groups", "0"], ["-row-groups", "1"], ["-row-groups", "1", "-atomic-rank-u", "1000"]):
        out = str(tmp_path / "o.vec")
        r = subprocess.run([os.path.join(ROOT, "word2bits"), "-train", corpus, "-output", out, "-threads", str(threads),
                            "-bitlevel", str(bitlevel), "-size", str(size), "-window", str(window),
                            "-negative", "0", "-iter", "2", "-min-count", "1", "-binary", "1", "-sample", "0",
                            "-positions", "53"] + flag, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-300:]
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] == outs[2]


@pytest.mark.parametrize("D,bitlevel", [(200, 1), (400, 2)])
def test_row_group_kernel_hogwild_matches_plain_kernel_statistically(gpu, D, bitlevel):
    """64 workers on shared rows (racy, as in the reference): same objective as the plain kernel -- epoch loss within 1 %,
    integer bookkeeping exact"""
    V, n, W = 20000, 2_000_000, 64
    rng = np.random.default_rng(17)
    ids = zipf_ids(rng, V, n).astype(np.int32)
    ids[1000::1000] = 0
    cn = counts_of(ids, V)
    starts = (np.arange(W, dtype=np.int64) * (n // W))
    res = [run(V, ids, cn, D, 8, 24, bitlevel, 4096, g, sample=0.0, threads=W, starts=starts) for g in (False, True)]
    assert (res[0][0], res[1][0]) == ("plain", "groups")
    assert res[0][3] == res[1][3]
    assert np.isfinite(res[1][1]).all() and np.isfinite(res[1][2]).all()
    print("GROUPS hogwild D=%d: plain %.1f groups %.1f (%.3f %%)" % (D, res[0][5], res[1][5], 100 * (res[1][5] / res[0][5] - 1)))
    assert res[1][5] == pytest.approx(res[0][5], rel=1e-2)


# ---------------------------------------------------------------------------------------------- against the ORACLE, directly
# (round-5 review, weak 1c: every test above compares HIP with HIP.  These two are the plain kernel's own oracle tests --
# tests/test_gpu_worker.py test_single_worker_short_horizon_tight / test_single_worker_long_horizon_statistical -- with the
# row-group kernel selected, at the BASELINE configs[0] / configs[2] row lengths where it is the automatic choice.)
def _oracle_setup(V, ids, D, window, negative, bitlevel, sample, iters, fma=False):
    from test_gpu_worker import setup
    return setup(V, ids, D, window, negative, bitlevel, sample, iters, fma=fma)


def _drift(a, b):
    return float(np.abs(a - b).mean()), float(np.mean(np.signbit(a) != np.signbit(b)))


@pytest.mark.parametrize("bitlevel,sample,D,window,negative", [
    (1, 1e-3, 200, 8, 24),     # configs[0] shape
    (2, 0.0, 400, 8, 24),      # configs[2] shape
    (0, 1e-3, 200, 8, 24),
    (1, 0.0, 800, 8, 24),      # configs[1] row length (explicit -row-groups 1)
    (1, 1e-3, 64, 5, 5),
])
def test_row_group_kernel_single_worker_short_horizon_against_oracle(gpu, bitlevel, sample, D, window, negative):
    """3000 positions over a 5000-word vocabulary: rows are rarely revisited, so the row-group kernel (producer wavefront:
    sentence reader, window and table draws, alpha; row groups: phases A / B / C) must track the CPU oracle closely --
    integer bookkeeping exact, values no farther from the bit-reference than 3x its own FMA build, epoch loss to 2e-3."""
    V, n = 5000, 3000
    rng = np.random.default_rng(4)
    ids = token_stream(rng, V, n)
    cn, tw, o = _oracle_setup(V, ids, D, window, negative, bitlevel, sample, 1)
    _, _, y = _oracle_setup(V, ids, D, window, negative, bitlevel, sample, 1, fma=True)
    t = w2b.Trainer(V, D, window, negative, bitlevel, num_threads=1, iter=1, sample=sample, train_words=tw, row_groups=True)
    t.set_model(o.u, o.v)
    t.set_vocab_counts(cn, 50000)
    t.set_corpus(ids)
    t.set_shards(np.zeros(1, np.int64))
    assert t.worker_kernel_name() == "groups"
    lo = o.train_epoch_tokens(ids, np.zeros(1, np.int64))
    y.train_epoch_tokens(ids, np.zeros(1, np.int64))
    lg = t.train_epoch(positions_per_launch=501)
    fin, wca, alpha, _ = t.epoch_status()
    assert fin and wca == o.m.word_count_actual and np.float32(alpha) == np.float32(o.m.alpha)
    u, v = t.get_model()
    for got, ref, yard in ((u, o.u, y.u), (v, o.v, y.v)):
        gm, gs = _drift(got, ref)
        ym, ys = _drift(yard, ref)
        floor = 1e-5 if bitlevel == 0 else 2e-3
        assert gm <= 3 * ym + floor, (gm, ym)
        assert gs <= 3 * ys + 2e-3, (gs, ys)
    assert lg == pytest.approx(lo, rel=2e-3)
    t.close()


@pytest.mark.parametrize("bitlevel,sample,D,window,negative", [
    (1, 1e-3, 200, 8, 24),
    (2, 0.0, 400, 8, 24),
    (0, 1e-2, 48, 3, 7),
])
def test_row_group_kernel_single_worker_long_horizon_against_oracle(gpu, bitlevel, sample, D, window, negative):
    """2 epochs x 30000 tokens on 150 rows (every row rewritten thousands of times; quantized training is chaotic: two builds
    of the REFERENCE disagree on 4-25 % of the signs here): integer bookkeeping exact, values no farther from the bit-reference
    than 1.5x the drift of the oracle's own FMA build, epoch loss to 2 %."""
    V, n = 150, 30000
    rng = np.random.default_rng(11)
    ids = token_stream(rng, V, n)
    ids[2000:3300] = zipf_ids(rng, V, 1300)
    cn, tw, o = _oracle_setup(V, ids, D, window, negative, bitlevel, sample, 2)
    _, _, y = _oracle_setup(V, ids, D, window, negative, bitlevel, sample, 2, fma=True)
    t = w2b.Trainer(V, D, window, negative, bitlevel, num_threads=1, iter=2, alpha=0.05, sample=sample, train_words=tw,
                    compute_loss=True, row_groups=True)
    t.set_model(o.u, o.v)
    t.set_vocab_counts(cn, 50000)
    t.set_corpus(ids)
    t.set_shards(np.zeros(1, np.int64))
    assert t.worker_kernel_name() == "groups"
    for ep in range(2):
        lo = o.train_epoch_tokens(ids, np.zeros(1, np.int64))
        y.train_epoch_tokens(ids, np.zeros(1, np.int64))
        lg = t.train_epoch(positions_per_launch=777)
        fin, wca, alpha, _ = t.epoch_status()
        assert fin and wca == o.m.word_count_actual and np.float32(alpha) == np.float32(o.m.alpha)
        assert lg == pytest.approx(lo, rel=2e-2)
    u, v = t.get_model()
    assert np.isfinite(u).all() and np.isfinite(v).all()
    for got, ref, yard in ((u, o.u, y.u), (v, o.v, y.v)):
        gm, gs = _drift(got, ref)
        ym, ys = _drift(yard, ref)
        assert gm <= 1.5 * ym + 1e-4, (gm, ym)
        assert gs <= 1.5 * ys + 1e-3, (gs, ys)
    t.close()


# ---------------------------------------------------------------------------------------------- refreshed read copies
@pytest.mark.parametrize("refresh", [4, 64])
def test_refreshed_copies_forced(gpu, refresh):
    """(advisor, round 5) the refreshed-copy path is the automatic default at 256 workers on Zipf vocabularies and had no test
    of its own: force refresh_rows_u = 4 / 64 against -1 (none) over several launches of 64 Hogwild workers -- integer
    bookkeeping exact, epoch loss within 1 % of the run without copies, and every launch ends in bounded time (the refresher
    exits with the workers instead of running into its time-out); 64 rows is the widest claim set the buffer holds."""
    import time
    V, n, W, D = 20000, 1_000_000, 64, 200
    rng = np.random.default_rng(23)
    ids = zipf_ids(rng, V, n).astype(np.int32)
    ids[1000::1000] = 0
    cn = counts_of(ids, V)
    starts = (np.arange(W, dtype=np.int64) * (n // W))
    t0 = time.time()
    base = run(V, ids, cn, D, 8, 24, 1, 1024, True, sample=0.0, threads=W, starts=starts, refresh_rows_u=-1, atomic_rank_u=2000)
    t_base = time.time() - t0
    t0 = time.time()
    got = run(V, ids, cn, D, 8, 24, 1, 1024, True, sample=0.0, threads=W, starts=starts, refresh_rows_u=refresh, atomic_rank_u=2000)
    t_got = time.time() - t0
    assert (base[0], got[0]) == ("groups", "groups")
    assert got[3] == base[3]                                   # integer bookkeeping exact
    assert np.isfinite(got[1]).all() and np.isfinite(got[2]).all()
    print("GROUPS refresh=%d: loss %.1f vs %.1f without (%.3f %%), %.1f s vs %.1f s" % (refresh, got[5], base[5], 100 * (got[5] / base[5] - 1), t_got, t_base))
    assert got[5] == pytest.approx(base[5], rel=1e-2)
    assert t_got < 5 * t_base + 10                             # no launch waited for the refresher's time-out
