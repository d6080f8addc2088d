"""not gpu: property test of the host-side ingest (word2bits_corpus.h) against the oracle's restatement of
ReadWord / LearnVocabFromTrainFile / the fseek shard arithmetic (ref src/word2bits.cpp:131-301,377) on random
byte soups: spaces, tabs, newlines, carriage returns (skipped, not separators), repeated separators, words that
are suffixes of other words, files that end without whitespace, empty files -- also with the multi-threaded
ingest forced to cut the file into up to 7 pieces (piece boundaries then fall next to every kind of separator)."""
import ctypes as C
import os

import numpy as np
from hypothesis import given, settings, strategies as st

import word2bits_amd as w2b
from w2b_testlib import oracle

ALPHABET = ["a", "b", "ab", "ba", "c", " ", " ", "\t", "\n", "\n", "\r"]


@settings(max_examples=120, deadline=None, derandomize=True, database=None)
@given(st.lists(st.sampled_from(ALPHABET), min_size=0, max_size=120), st.integers(1, 3), st.integers(1, 9),
       st.sampled_from([None, (7, 1), (3, 5), (2, 40)]))
def test_ingest_matches_oracle_on_random_text(tmp_path_factory, pieces, min_count, nthreads, host_split):
    text = "".join(pieces)
    d = tmp_path_factory.mktemp("fz")
    p = str(d / "c.txt")
    with open(p, "w", newline="") as f:
        f.write(text)
    O = oracle()
    vb = O.w2bo_vocab_learn(p.encode(), min_count)
    # host_split = (threads, min piece bytes): force the parallel ingest to cut even these tiny files into pieces
    for k in ("W2B_INGEST_THREADS", "W2B_INGEST_MIN_PIECE"):
        os.environ.pop(k, None)
    if host_split:
        os.environ["W2B_INGEST_THREADS"], os.environ["W2B_INGEST_MIN_PIECE"] = str(host_split[0]), str(host_split[1])
    try:
        c = w2b.Corpus(p, min_count)
    finally:
        for k in ("W2B_INGEST_THREADS", "W2B_INGEST_MIN_PIECE"):
            os.environ.pop(k, None)
    try:
        V = O.w2bo_vocab_size(vb)
        assert c.vocab_size == V
        assert c.train_words == O.w2bo_vocab_train_words(vb)
        assert c.file_size == len(text.encode())
        assert c.words() == [O.w2bo_vocab_word(vb, i).decode() for i in range(V)]
        assert c.counts().tolist() == [O.w2bo_vocab_count(vb, i) for i in range(V)]
        ids, bg = C.POINTER(C.c_int)(), C.POINTER(C.c_longlong)()
        n = O.w2bo_tokenize_file(vb, p.encode(), C.byref(ids), C.byref(bg))
        oid = np.ctypeslib.as_array(ids, shape=(n,)).copy() if n > 0 else np.zeros(0, np.int32)
        assert np.array_equal(oid[oid >= 0], c.tokens())
        st_, ov = c.shards(nthreads)
        for w in range(nthreads):
            o = C.c_int(0)
            s = O.w2bo_shard_start(vb, p.encode(), c.file_size // nthreads * w, bg, n, C.byref(o))
            assert int((oid[:s] >= 0).sum()) == st_[w], (text, w)
            assert o.value == ov[w], (text, w, o.value, ov[w])
    finally:
        O.w2bo_vocab_free(vb)
        c.close()
