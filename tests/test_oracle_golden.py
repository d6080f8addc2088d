"""not gpu: pins the CPU oracle (oracle/w2b_oracle.c) against
  (1) committed golden output files produced by the UNMODIFIED reference program
      (tests/golden/*.vec, generator tests/golden/make_golden.py), byte for byte;
  (2) the reference binary itself where oracle/_ref/ exists (built from /root/reference);
  (3) the known answers printed in the reference's README (README.md:12-17,122-131).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from w2b_testlib import GOLDEN, oracle, ref_binary, run_ref, write_corpus, read_vectors

META = json.load(open(os.path.join(GOLDEN, "golden.json")))
CORPUS = os.path.join(GOLDEN, "corpus_small.txt")


def run_oracle(train, out, flags, threads=1):
    f = dict(bitlevel=1, size=100, window=5, negative=5, iter=5, min_count=5, alpha=0.05, sample=1e-3,
             reg=0.0, binary=0)
    f.update(flags)
    losses = (C.c_double * max(1, f["iter"]))()
    rc = oracle().w2bo_run(train.encode(), out.encode(), f["bitlevel"], f["size"], f["window"],
                           f["negative"], threads, f["iter"], f["min_count"], f["alpha"], f["sample"],
                           f["reg"], f["binary"], losses)
    assert rc == 0
    return list(losses)[:f["iter"]]


@pytest.mark.parametrize("name", sorted(META))
def test_oracle_reproduces_reference_output_bytes(name, tmp_path):
    flags = META[name]["flags"]
    out = str(tmp_path / "o.vec")
    losses = run_oracle(CORPUS, out, flags)
    want = open(os.path.join(GOLDEN, name + ".vec"), "rb").read()
    got = open(out, "rb").read()
    assert got == want                                   # whole file, incl. header and -0.0 bits
    for a, b in zip(losses, META[name]["epoch_loss"]):   # "Epoch Loss: %lf" of the reference
        assert abs(a - b) <= 5e-7 * max(1.0, abs(b)) + 1e-6


def test_oracle_vocab_matches_reference_counts():
    L = oracle()
    for name, m in META.items():
        vb = L.w2bo_vocab_learn(CORPUS.encode(), m["flags"]["min_count"])
        assert L.w2bo_vocab_size(vb) == m["vocab_size"]
        assert L.w2bo_vocab_train_words(vb) == m["train_words"]
        L.w2bo_vocab_free(vb)


@pytest.mark.skipif(ref_binary("word2bits_nofma") is None, reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("flags", [
    dict(bitlevel=1, size=20, window=8, negative=24, iter=2, min_count=5, binary=1),
    dict(bitlevel=2, size=24, window=5, negative=5, iter=1, min_count=2, binary=1, sample=1e-2),
    dict(bitlevel=0, size=16, window=3, negative=2, iter=3, min_count=1, binary=0, alpha=0.025),
])
def test_oracle_vs_live_reference_binary(flags, tmp_path):
    corpus = write_corpus(str(tmp_path / "c.txt"), seed=3, vocab=200, n_tokens=12000)
    ref_out, ora_out = str(tmp_path / "r.vec"), str(tmp_path / "o.vec")
    run_ref("word2bits_nofma", corpus, ref_out, threads=1, **flags)
    run_oracle(corpus, ora_out, flags)
    assert open(ref_out, "rb").read() == open(ora_out, "rb").read()


def test_quantize_truth_table():
    """ref src/word2bits.cpp:73-108; 1-bit levels are +-0x3EAAAAAB (README.md:125-131)."""
    q = oracle().w2bo_quantize
    bits = lambda x: np.float32(x).view(np.uint32)
    for x in (0.7, 1e-30, 0.0, -0.0, float("nan"), 3.0):
        assert bits(q(x, 1)) == 0x3EAAAAAB
    for x in (-0.7, -1e-30, -3.0):
        assert bits(q(x, 1)) == 0xBEAAAAAB
    assert q(0.123, 0) == np.float32(0.123)
    assert [q(x, 2) for x in (0.0, 0.5, 0.50001, 2.0, -0.5, -0.51)] == [.25, .25, .75, .75, -.25, -.75]
    assert bits(q(float("nan"), 2)) == bits(0.75)
    assert bits(q(0.3, 3)) == bits(0.0) and bits(q(-0.3, 3)) == bits(-0.0)      # degenerate bitlevel
    assert [q(x, 4) for x in (0.0, 0.06, 0.07, 0.99, 5.0, -0.31)] == [0.0, 0.0, 0.125, 1.0, 1.0, -0.25]
    assert q(0.5, 8) == 0.5 and q(0.501, 8) == 0.5 and q(0.505, 8) == np.float32(65 / 128)


def test_exp_table_and_index():
    L = oracle()
    t = np.zeros(1001, np.float32)
    L.w2bo_build_exp_table(t.ctypes.data_as(C.POINTER(C.c_float)))
    assert abs(t[500] - 0.5) < 1e-6 and t[0] < 0.0025 and t[999] > 0.9974
    assert np.all(np.diff(t[:1000]) > 0)
    # EXP_TABLE_SIZE / MAX_EXP / 2 is integer arithmetic = 83 (ref :475): max index 996, not 999
    assert L.w2bo_exp_index(6.0) == 996 and L.w2bo_exp_index(-6.0) == 0 and L.w2bo_exp_index(0.0) == 498


def test_init_net_is_the_readme_pattern():
    """InitNet (ref :343-361): v first then u, values k/65536-0.5, period 65536 in the draw index."""
    L = oracle()
    V, D = 700, 100
    u, v = np.zeros((V, D), np.float32), np.zeros((V, D), np.float32)
    L.w2bo_init_net(V, D, u.ctypes.data_as(C.POINTER(C.c_float)), v.ctypes.data_as(C.POINTER(C.c_float)))
    s, vals = 1, []
    for _ in range(5):
        s = (s * 25214903917 + 11) % (1 << 64)
        vals.append(np.float32((s & 0xFFFF) / 65536.0 - 0.5))
    assert v.ravel()[:5].tolist() == vals
    flat = np.concatenate([v.ravel(), u.ravel()])
    assert np.array_equal(flat[:4464], flat[65536:65536 + 4464])
    assert np.abs(flat).max() <= 0.5


@pytest.mark.skipif(ref_binary("word2bits_nofma") is None, reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed", range(16))
def test_oracle_vs_live_reference_random_flags(seed, tmp_path):
    """a seeded sweep over the flag space (all bitlevels incl. 3, -size 1..40, -window 1..10, -negative 0..12,
    -sample on/off, -reg, -alpha, 1..3 epochs, text and binary, with and without the tokenizer's corner cases):
    the oracle's output file equals the live reference's, byte for byte"""
    rng = np.random.default_rng(1000 + seed)
    flags = dict(bitlevel=int(rng.choice([0, 1, 1, 2, 3, 4, 8])), size=int(rng.integers(1, 41)),
                 window=int(rng.integers(1, 11)), negative=int(rng.integers(0, 13)), iter=int(rng.integers(1, 4)),
                 min_count=int(rng.integers(1, 4)), binary=int(rng.integers(0, 2)),
                 sample=float(rng.choice([0.0, 1e-3, 1e-2, 0.1])), reg=float(rng.choice([0.0, 0.0, 1e-3])),
                 alpha=float(rng.choice([0.05, 0.025, 0.1])))
    corpus = write_corpus(str(tmp_path / "c.txt"), seed=seed, vocab=int(rng.integers(20, 400)),
                          n_tokens=int(rng.integers(2000, 9000)), line_len=int(rng.integers(3, 60)),
                          quirks=bool(rng.integers(0, 2)))
    ref_out, ora_out = str(tmp_path / "r.vec"), str(tmp_path / "o.vec")
    run_ref("word2bits_nofma", corpus, ref_out, threads=1, **flags)
    run_oracle(corpus, ora_out, flags)
    assert open(ref_out, "rb").read() == open(ora_out, "rb").read(), flags


@pytest.mark.skipif(ref_binary("word2bits_nofma") is None, reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("threads,prefix,flags", [
    (4, b"", dict(bitlevel=1, size=12, window=5, iter=1, sample=0.0)),
    (4, b"", dict(bitlevel=0, size=9, window=3, iter=2, sample=1e-2)),
    (3, b"zz zz\n", dict(bitlevel=2, size=10, window=8, iter=2, sample=0.0)),      # thread offsets land inside words
    (7, b"q\n", dict(bitlevel=1, size=8, window=2, iter=1, sample=1e-3)),
])
def test_multi_thread_shard_logic_vs_live_reference(threads, prefix, flags, tmp_path):
    """-threads N on a corpus whose shards use disjoint vocabularies, with -negative 0 and shards shorter than one
    alpha period: no two threads touch the same row, so the reference itself is deterministic and the oracle's
    per-thread logic (offsets, mid-word starts, quotas, seeds) must reproduce its file byte for byte."""
    from w2b_testlib import write_disjoint_shard_corpus
    corpus = write_disjoint_shard_corpus(str(tmp_path / "c.txt"), n_shards=threads, seed=threads, prefix=prefix)
    f = dict(negative=0, min_count=1, binary=1)
    f.update(flags)
    outs = []
    for rep in range(2):                                   # the reference really is deterministic here
        o = str(tmp_path / ("r%d.vec" % rep))
        run_ref("word2bits_nofma", corpus, o, threads=threads, **f)
        outs.append(open(o, "rb").read())
    assert outs[0] == outs[1]
    ora = str(tmp_path / "o.vec")
    run_oracle(corpus, ora, f, threads=threads)
    assert open(ora, "rb").read() == outs[0]
