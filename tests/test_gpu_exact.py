"""-m gpu: the parity mode (w2b_config.exact_reduction / ./word2bits -exact 1).  With the dot product accumulated
in the reference's own order the HIP path has no re-association left, so everything is compared BIT-EXACTLY:
single updates against the oracle, whole single-worker epochs (sentence reader, sub-sampling, window and negative
draws, alpha staircase, quantized training -- chaotic, so any one-ulp slip would show) against the oracle, and the
command line's output FILES against the committed files written by the unmodified reference program."""
import json
import os
import subprocess

import numpy as np
import pytest

import word2bits_amd as w2b
from w2b_testlib import GOLDEN, ROOT, OracleState
from test_gpu_parity import disjoint_tuples
from test_gpu_worker import token_stream, setup

pytestmark = pytest.mark.gpu
META = json.load(open(os.path.join(GOLDEN, "golden.json")))


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.mark.parametrize("D,window,negative,bitlevel,reg", [
    (800, 8, 24, 1, 0.0), (200, 8, 24, 1, 0.0), (400, 8, 24, 2, 0.0), (1000, 5, 12, 0, 0.0), (100, 5, 5, 4, 0.001),
    (64, 3, 7, 8, 0.0), (50, 4, 3, 1, 0.0), (1200, 2, 3, 2, 0.0), (36, 40, 70, 1, 0.0), (8, 1, 0, 0, 0.0),
    (257, 3, 5, 0, 0.0), (1, 3, 2, 2, 0.0),
    # rows longer than a workgroup has columns (process_word_wide; the reference has no limit on -size, ref :598)
    (5000, 3, 5, 1, 0.0), (1030, 4, 6, 2, 0.001), (4100, 2, 3, 0, 0.0),
])
def test_tuple_updates_bit_exact(gpu, D, window, negative, bitlevel, reg):
    n = 24
    V = n * (2 * window + negative + 2) + 64
    rng = np.random.default_rng(D)
    cn = np.concatenate([[0], np.sort(rng.integers(5, 2000, V - 1))[::-1]]).astype(np.int64)
    o = OracleState(cn, D, window=window, negative=negative, bitlevel=bitlevel, reg=reg, sample=0.0, table_size=20000)
    o.u *= 3.0
    o.v *= 3.0
    t = w2b.Trainer(V, D, window, negative, bitlevel, num_threads=1, iter=1, alpha=0.05, sample=0.0, reg=reg,
                    train_words=int(cn.sum()), compute_loss=False, exact=True)
    t.set_model(o.u, o.v)
    center, ctx_off, ctx, neg = disjoint_tuples(rng, V, n, window, negative)
    for rep in range(3):                       # the same rows again: errors would compound
        o.train_tuples(center, ctx_off, ctx, neg, 0.05)
        t.train_tuples(center, ctx_off, ctx, neg, 0.05, serial=False)
    u, v = t.get_model()
    assert same_bits(u, o.u) and same_bits(v, o.v), (np.abs(u - o.u).max(), np.abs(v - o.v).max())
    t.close()


@pytest.mark.parametrize("bitlevel,sample,D,window,negative,iters", [
    (1, 1e-3, 200, 8, 24, 2), (0, 1e-3, 200, 8, 24, 1), (2, 0.0, 100, 3, 7, 2), (1, 1e-3, 32, 1, 0, 3),
    (4, 1e-3, 50, 5, 5, 2), (1, 0.0, 800, 8, 24, 1),
    (1, 1e-3, 5000, 3, 4, 1), (2, 0.0, 1030, 2, 3, 1),           # wide rows (process_word_wide)
])
def test_single_worker_epochs_bit_exact(gpu, bitlevel, sample, D, window, negative, iters):
    """60 000 tokens over 150 words: every row is revisited hundreds of times, quantized training is chaotic (two
    builds of the reference drift apart by percents here), so bit equality after whole epochs means every single
    update was bit-exact."""
    V, n = 150, (60000 if D <= 1024 else 6000)
    rng = np.random.default_rng(9)
    ids = token_stream(rng, V, n)
    cn, tw, o = setup(V, ids, D, window, negative, bitlevel, sample, iters)
    t = w2b.Trainer(V, D, window, negative, bitlevel, num_threads=1, iter=iters, sample=sample, train_words=tw,
                    compute_loss=False, exact=True)
    t.set_model(o.u, o.v)
    t.set_vocab_counts(cn, 50000)
    t.set_corpus(ids)
    t.set_shards(np.zeros(1, np.int64))
    for it in range(iters):
        o.train_epoch_tokens(ids, np.zeros(1, np.int64))
        t.train_epoch(positions_per_launch=777)
        fin, wca, alpha, _ = t.epoch_status()
        assert fin and wca == o.m.word_count_actual and np.float32(alpha) == np.float32(o.m.alpha)
        u, v = t.get_model()
        assert same_bits(u, o.u) and same_bits(v, o.v), (it, np.abs(u - o.u).max(), np.abs(v - o.v).max())
    t.close()


@pytest.mark.parametrize("name", sorted(META))
def test_cli_exact_output_file_is_byte_identical_to_reference(gpu, name, tmp_path):
    """./word2bits -threads 1 -exact 1 with the flags of every committed golden run (bitlevel 0/1/2/4/8, -reg,
    sub-sampling on and off, text and binary): the output file equals the unmodified reference's, byte for byte --
    vocabulary, InitNet, every training update of every epoch, quantize(u+v), writer."""
    out = str(tmp_path / "o.vec")
    args = [os.path.join(ROOT, "word2bits"), "-train", os.path.join(GOLDEN, "corpus_small.txt"), "-output", out,
            "-threads", "1", "-exact", "1"]
    for k, v in META[name]["flags"].items():
        args += ["-" + k.replace("_", "-"), str(v)]
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-300:] + r.stderr[-300:]
    assert open(out, "rb").read() == open(os.path.join(GOLDEN, name + ".vec"), "rb").read()


@pytest.mark.parametrize("seed", range(10))
def test_cli_exact_equals_oracle_on_random_flags(gpu, seed, tmp_path):
    """the same seeded sweep over the flag space that pins the oracle to the live reference on the CPU side
    (tests/test_oracle_golden.py): ./word2bits -threads 1 -exact 1 writes the oracle's file, byte for byte"""
    from w2b_testlib import write_corpus
    from test_oracle_golden import run_oracle
    rng = np.random.default_rng(1000 + seed)
    flags = dict(bitlevel=int(rng.choice([0, 1, 1, 2, 3, 4, 8])), size=int(rng.integers(1, 41)),
                 window=int(rng.integers(1, 11)), negative=int(rng.integers(0, 13)), iter=int(rng.integers(1, 4)),
                 min_count=int(rng.integers(1, 4)), binary=int(rng.integers(0, 2)),
                 sample=float(rng.choice([0.0, 1e-3, 1e-2, 0.1])), reg=float(rng.choice([0.0, 0.0, 1e-3])),
                 alpha=float(rng.choice([0.05, 0.025, 0.1])))
    corpus = write_corpus(str(tmp_path / "c.txt"), seed=seed, vocab=int(rng.integers(20, 400)),
                          n_tokens=int(rng.integers(2000, 9000)), line_len=int(rng.integers(3, 60)),
                          quirks=bool(rng.integers(0, 2)))
    gpu_out, ora_out = str(tmp_path / "g.vec"), str(tmp_path / "o.vec")
    args = [os.path.join(ROOT, "word2bits"), "-train", corpus, "-output", gpu_out, "-threads", "1", "-exact", "1"]
    for k, v in flags.items():
        args += ["-" + k.replace("_", "-"), repr(v) if isinstance(v, float) else str(v)]
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-300:] + r.stderr[-300:]
    run_oracle(corpus, ora_out, flags)
    assert open(gpu_out, "rb").read() == open(ora_out, "rb").read(), flags


@pytest.mark.parametrize("threads,prefix,flags", [
    (4, b"", dict(bitlevel=1, size=12, window=5, iter=1, sample=0.0)),
    (4, b"", dict(bitlevel=0, size=9, window=3, iter=2, sample=1e-2)),
    (3, b"zz zz\n", dict(bitlevel=2, size=10, window=8, iter=2, sample=0.0)),      # worker starts inside words
    (7, b"q\n", dict(bitlevel=1, size=8, window=2, iter=1, sample=1e-3)),
    (16, b"", dict(bitlevel=1, size=200, window=8, iter=2, sample=0.0)),
])
def test_multi_worker_shard_logic_bit_exact(gpu, threads, prefix, flags, tmp_path):
    """Hogwild with SEVERAL workers, still bit-exact: shards with disjoint vocabularies, -negative 0, shards shorter
    than an alpha period -- no two workers touch the same row, so the run is deterministic (the reference's own
    -threads N run is, and the oracle is pinned to it on exactly these corpora, tests/test_oracle_golden.py).
    Checks worker ids as seeds, shard offsets incl. mid-word starts, per-worker quotas and the sentence that crosses
    the quota (read, not trained) through the command line, in parity mode."""
    from w2b_testlib import write_disjoint_shard_corpus
    from test_oracle_golden import run_oracle
    corpus = write_disjoint_shard_corpus(str(tmp_path / "c.txt"), n_shards=threads, seed=threads, prefix=prefix)
    f = dict(negative=0, min_count=1, binary=1)
    f.update(flags)
    gpu_out, ora_out = str(tmp_path / "g.vec"), str(tmp_path / "o.vec")
    args = [os.path.join(ROOT, "word2bits"), "-train", corpus, "-output", gpu_out, "-threads", str(threads),
            "-exact", "1", "-positions", "37"]
    for k, v in f.items():
        args += ["-" + k.replace("_", "-"), repr(v) if isinstance(v, float) else str(v)]
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-300:] + r.stderr[-300:]
    run_oracle(corpus, ora_out, f, threads=threads)
    assert open(gpu_out, "rb").read() == open(ora_out, "rb").read()
