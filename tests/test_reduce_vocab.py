"""not gpu: ReduceVocab (ref src/word2bits.cpp:245-263, called at :293 while the vocabulary is learned).

The reference calls it whenever its table holds more than 0.7 x vocab_hash_size words -- 21 M with the constant of
ref :35, which no test can afford.  oracle/Makefile therefore also builds the reference with that ONE constant changed
to 3000 on its way into the compiler (oracle/_ref/word2bits_nofma_hash3000); what that binary does on two corpora
generated from integers is committed as tests/golden/reduce_vocab.json (generator make_reduce_vocab_golden.py).

  (1) the oracle's restatement (w2bo_vocab_learn_ex / w2bo_run_ex) reproduces the fixture: vocabulary size, word count,
      vocabulary order, and the sha256 of the complete output file of a training run -- i.e. every count, since the
      unigram table and the sub-sampling are built from them;
  (2) the same against the live binary on other corpora, where oracle/_ref exists;
  (3) the product's ingest (w2b_corpus_load_ex: parallel tokeniser + sequential replay) equals the oracle: words,
      counts, token stream, shard starts -- also when the file is cut into many pieces.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import word2bits_amd as w2b
from w2b_testlib import GOLDEN, oracle, ref_binary, run_ref, write_reduce_vocab_corpus

META = json.load(open(os.path.join(GOLDEN, "reduce_vocab.json")))
HASH = 3000


def sha(b):
    return hashlib.sha256(b).hexdigest()


def oracle_run(corpus, out, flags, hash_size=HASH):
    f = dict(alpha=0.05, sample=1e-3, reg=0.0)
    f.update(flags)
    losses = (C.c_double * max(1, f["iter"]))()
    rc = oracle().w2bo_run_ex(corpus.encode(), out.encode(), f["bitlevel"], f["size"], f["window"], f["negative"], 1,
                              f["iter"], f["min_count"], f["alpha"], f["sample"], f["reg"], f["binary"], losses,
                              hash_size)
    assert rc == 0


def oracle_vocab(corpus, min_count, hash_size=HASH):
    O = oracle()
    vb = O.w2bo_vocab_learn_ex(corpus.encode(), min_count, hash_size)
    n = O.w2bo_vocab_size(vb)
    words = [O.w2bo_vocab_word(vb, i).decode() for i in range(n)]
    counts = [O.w2bo_vocab_count(vb, i) for i in range(n)]
    return vb, words, counts


@pytest.mark.parametrize("kind", ["zipf", "wipe"])
def test_oracle_reproduces_the_reference_with_a_small_hash_table(kind, tmp_path):
    corpus = write_reduce_vocab_corpus(str(tmp_path / "c.txt"), kind)
    assert sha(open(corpus, "rb").read()) == META[kind]["corpus_sha256"]
    for name in ("iter0", "train"):
        m = META[kind][name]
        vb, words, counts = oracle_vocab(corpus, m["flags"]["min_count"])
        assert len(words) == m["vocab_size"] and sum(counts) == m["train_words"]
        assert oracle().w2bo_vocab_train_words(vb) == m["train_words"]
        assert words[0] == m["row0"] and sha("\n".join(words).encode("latin1")) == m["words_sha256"]
        oracle().w2bo_vocab_free(vb)
        out = str(tmp_path / "o.vec")
        oracle_run(corpus, out, m["flags"])
        assert sha(open(out, "rb").read()) == m["output_sha256"]
    # the corpora do what they were built for: the reduction really ran (the full table would be larger) ...
    vb, full, _ = oracle_vocab(corpus, 1, 30000000)
    oracle().w2bo_vocab_free(vb)
    assert len(full) > 0.7 * HASH > META[kind]["iter0"]["vocab_size"]
    # ... and on "wipe" it took "</s>" with it: another word owns row 0, "</s>" is back as an ordinary word
    if kind == "wipe":
        vb, words, _ = oracle_vocab(corpus, 1)
        oracle().w2bo_vocab_free(vb)
        assert words[0] != "</s>" and "</s>" in words[1:]


@pytest.mark.skipif(ref_binary("word2bits_nofma_hash3000") is None, reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("kind,seed", [("zipf", 1), ("wipe", 2), ("zipf", 3)])
def test_oracle_vs_live_reference_with_a_small_hash_table(kind, seed, tmp_path):
    corpus = write_reduce_vocab_corpus(str(tmp_path / "c.txt"), kind, seed=seed)
    flags = dict(bitlevel=1, size=6, window=4, negative=3, iter=1, min_count=2, binary=1)
    ref_out, ora_out = str(tmp_path / "r.vec"), str(tmp_path / "o.vec")
    run_ref("word2bits_nofma_hash3000", corpus, ref_out, threads=1, **flags)
    oracle_run(corpus, ora_out, flags)
    assert open(ref_out, "rb").read() == open(ora_out, "rb").read()


@pytest.mark.parametrize("host_split", [None, (7, 1), (64, 64)])
@pytest.mark.parametrize("kind", ["zipf", "wipe"])
def test_ingest_with_reduce_vocab_matches_oracle(kind, host_split, tmp_path, monkeypatch):
    if host_split:   # force the parallel tokeniser to cut the file into pieces: the replay must still be sequential
        monkeypatch.setenv("W2B_INGEST_THREADS", str(host_split[0]))
        monkeypatch.setenv("W2B_INGEST_MIN_PIECE", str(host_split[1]))
    corpus = write_reduce_vocab_corpus(str(tmp_path / "c.txt"), kind)
    O = oracle()
    for min_count in (1, 3):
        c = w2b.Corpus(corpus, min_count, vocab_hash_size=HASH)
        vb, words, counts = oracle_vocab(corpus, min_count)
        assert c.words() == words and c.counts().tolist() == counts
        assert c.train_words == O.w2bo_vocab_train_words(vb)
        ids, bg = C.POINTER(C.c_int)(), C.POINTER(C.c_longlong)()
        n = O.w2bo_tokenize_file(vb, corpus.encode(), C.byref(ids), C.byref(bg))
        oid = np.ctypeslib.as_array(ids, shape=(n,)).copy()
        assert np.array_equal(oid[oid >= 0], c.tokens())
        for nt in (1, 3, 8):
            st, ov = c.shards(nt)
            for w in range(nt):
                o = C.c_int(0)
                s = O.w2bo_shard_start(vb, corpus.encode(), c.file_size // nt * w, bg, n, C.byref(o))
                assert int((oid[:s] >= 0).sum()) == st[w] and o.value == ov[w]
        O.w2bo_vocab_free(vb)
        c.close()
    m = META[kind]["iter0"]                              # and the reference's own numbers
    c = w2b.Corpus(corpus, 1, vocab_hash_size=HASH)
    assert (c.vocab_size, c.train_words, c.words()[0]) == (m["vocab_size"], m["train_words"], m["row0"])
    assert sha("\n".join(c.words()).encode("latin1")) == m["words_sha256"]
    c.close()


def test_default_hash_size_never_reduces_these_corpora(tmp_path):
    corpus = write_reduce_vocab_corpus(str(tmp_path / "c.txt"), "wipe")
    a, b = w2b.Corpus(corpus, 1), w2b.Corpus(corpus, 1, vocab_hash_size=30000000)
    assert a.words() == b.words() and a.words()[0] == "</s>" and a.vocab_size > 0.7 * HASH
    assert np.array_equal(a.tokens(), b.tokens())
    a.close()
    b.close()
