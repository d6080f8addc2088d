"""not gpu: host-side product code (corpus ingest, tables, C-ABI surface, CLI argument handling)
against the CPU oracle and the committed reference fixtures.  No compute entry point is called."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import word2bits_amd as w2b
from word2bits_amd import _lib
from w2b_testlib import GOLDEN, ROOT, oracle, write_corpus, read_vectors, fptr, iptr, lptr

META = json.load(open(os.path.join(GOLDEN, "golden.json")))
CORPUS = os.path.join(GOLDEN, "corpus_small.txt")


def test_library_exports_every_declared_symbol():
    L = w2b.lib()
    declared = set()
    for h in ("word2bits_hip.h", "word2bits_corpus.h", "word2bits_eval.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        declared |= set(re.findall(r"\b(w2b_[a-z0-9_]+)\s*\(", src))
    assert len(declared) >= 50
    for name in declared:
        assert hasattr(L, name), "library does not export " + name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_no_cpu_fallback_without_gpu():
    if w2b.lib().w2b_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(w2b.W2bError) as e:
        w2b.Trainer(100, 8)
    assert e.value.code == _lib.W2B_ENOGPU


def test_config_struct_layout_matches_header():
    assert C.sizeof(_lib.Config) == 3 * 8 + 5 * 4 + 3 * 4 + 2 * 4 + 2 * 4 + 5 * 4 + 4  # incl. tail padding


@pytest.mark.parametrize("min_count", [1, 2, 5])
@pytest.mark.parametrize("host_split", [None, (8, 1000), (64, 64)])
def test_corpus_ingest_matches_oracle_and_reference(min_count, host_split, tmp_path, monkeypatch):
    if host_split:   # force the parallel ingest to cut the small test files into pieces
        monkeypatch.setenv("W2B_INGEST_THREADS", str(host_split[0]))
        monkeypatch.setenv("W2B_INGEST_MIN_PIECE", str(host_split[1]))
    O = oracle()
    for path in (CORPUS, write_corpus(str(tmp_path / "c.txt"), seed=5, vocab=400, n_tokens=20000)):
        c = w2b.Corpus(path, min_count)
        vb = O.w2bo_vocab_learn(path.encode(), min_count)
        assert c.vocab_size == O.w2bo_vocab_size(vb)
        assert c.train_words == O.w2bo_vocab_train_words(vb)
        assert c.file_size == O.w2bo_vocab_file_size(vb) == os.path.getsize(path)
        assert c.words() == [O.w2bo_vocab_word(vb, i).decode() for i in range(c.vocab_size)]
        assert c.counts().tolist() == [O.w2bo_vocab_count(vb, i) for i in range(c.vocab_size)]
        ids, bg = C.POINTER(C.c_int)(), C.POINTER(C.c_longlong)()
        n = O.w2bo_tokenize_file(vb, path.encode(), C.byref(ids), C.byref(bg))
        oid = np.ctypeslib.as_array(ids, shape=(n,)).copy()
        assert np.array_equal(oid[oid >= 0], c.tokens())
        for nt in (1, 2, 3, 5, 12, 31):
            st, ov = c.shards(nt)
            for w in range(nt):
                o = C.c_int(0)
                s = O.w2bo_shard_start(vb, path.encode(), c.file_size // nt * w, bg, n, C.byref(o))
                assert int((oid[:s] >= 0).sum()) == st[w] and o.value == ov[w]
        O.w2bo_vocab_free(vb)
        c.close()
    # against the reference's own stdout (golden.json)
    for m in META.values():
        if m["flags"]["min_count"] == min_count:
            c = w2b.Corpus(CORPUS, min_count)
            assert (c.vocab_size, c.train_words) == (m["vocab_size"], m["train_words"])
            c.close()


def test_shard_seek_inside_a_word_that_is_itself_a_word(tmp_path):
    """ref :377: fseek lands mid-word; the tail may be a vocabulary word ('xab' -> 'ab')."""
    p = str(tmp_path / "s.txt")
    open(p, "w").write("ab xab ab ab\nxab b ab xab b b\n")
    c = w2b.Corpus(p, 1)
    assert c.words()[0] == "</s>" and set(c.words()[1:]) == {"ab", "xab", "b"}
    for off, expect in ((4, "ab"), (5, "b"), (3, None), (2, None)):
        # emulate num_threads so that file_size // nt * w == off is not needed: use the oracle helper
        vb = oracle().w2bo_vocab_learn(p.encode(), 1)
        ids, bg = C.POINTER(C.c_int)(), C.POINTER(C.c_longlong)()
        n = oracle().w2bo_tokenize_file(vb, p.encode(), C.byref(ids), C.byref(bg))
        o = C.c_int(0)
        oracle().w2bo_shard_start(vb, p.encode(), off, bg, n, C.byref(o))
        assert o.value == (-2 if expect is None else c.search(expect))
    # the product's shard arithmetic on the same file: 30 bytes / 6 threads = offsets 0,5,10,...
    st, ov = c.shards(6)
    assert ov[1] == c.search("b") and st[0] == 0
    c.close()


def test_host_tables_match_oracle():
    L, O = w2b.lib(), oracle()
    a, b = np.zeros(1000, np.float32), np.zeros(1001, np.float32)
    L.w2b_build_exp_table(a.ctypes.data_as(_lib.f32p))
    O.w2bo_build_exp_table(fptr(b))
    assert np.array_equal(a.view(np.uint32), b[:1000].view(np.uint32))
    rng = np.random.default_rng(0)
    cn = np.concatenate([[3], np.sort(rng.integers(5, 9000, 999))[::-1]]).astype(np.int64)
    t1, t2 = np.zeros(200000, np.int32), np.zeros(200000, np.int32)
    assert L.w2b_build_unigram_table(cn.ctypes.data_as(_lib.i64p), len(cn), t1.ctypes.data_as(_lib.i32p), len(t1)) == 0
    O.w2bo_build_unigram_table(lptr(cn), len(cn), iptr(t2), len(t2))
    assert np.array_equal(t1, t2)
    k1 = np.zeros(len(cn), np.float32)
    L.w2b_build_keep_prob(cn.ctypes.data_as(_lib.i64p), len(cn), 1e-3, int(cn.sum()), k1.ctypes.data_as(_lib.f32p))
    k2 = np.array([O.w2bo_keep_prob(int(c), 1e-3, int(cn.sum())) for c in cn], np.float32)
    assert np.array_equal(k1.view(np.uint32), k2.view(np.uint32))
    xs = np.concatenate([rng.standard_normal(2000).astype(np.float32), np.float32([0, -0.0, .5, -.5, 1, 7, np.nan])])
    for bl in (0, 1, 2, 3, 4, 5, 8, 16):
        got = np.array([L.w2b_quantize(float(x), bl) for x in xs], np.float32)
        want = np.array([O.w2bo_quantize(float(x), bl) for x in xs], np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), bl


@pytest.mark.parametrize("name", ["b1_iter0", "b0_iter0_text", "b2_d10_text"])
def test_save_vectors_reproduces_reference_file_format(name, tmp_path):
    """Writer of ref :560-576: feed it the reference's own values, expect the reference's own bytes."""
    flags = META[name]["flags"]
    golden = os.path.join(GOLDEN, name + ".vec")
    words, M = read_vectors(golden, flags["binary"])
    c = w2b.Corpus(CORPUS, flags["min_count"])
    assert c.words() == words
    out = str(tmp_path / "o.vec")
    c.save_vectors(out, M, flags["binary"])
    assert open(out, "rb").read() == open(golden, "rb").read()
    c.close()


CLI = os.path.join(ROOT, "word2bits")


@pytest.mark.skipif(not os.path.exists(CLI), reason="CLI not built")
def test_cli_argument_handling_like_reference(tmp_path):
    # a flag in last position without value (ref :582-585)
    r = subprocess.run([CLI, "-train", CORPUS, "-size"], capture_output=True, text=True)
    assert r.returncode == 1 and "Argument missing for -size" in r.stdout
    # missing training file (ref :271-274)
    r = subprocess.run([CLI, "-train", str(tmp_path / "nope.txt"), "-output", str(tmp_path / "o")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "ERROR: training data file not found!" in r.stdout
    # no -output: vocabulary statistics only (ref :527), works without a GPU
    r = subprocess.run([CLI, "-train", CORPUS, "-min-count", "2", "-bogus-flag", "1"], capture_output=True, text=True)
    m = META["b1_d8"]
    assert r.returncode == 0
    assert "Starting training using file %s" % CORPUS in r.stdout
    assert "Vocab size: %d" % m["vocab_size"] in r.stdout
    assert "Words in train file: %d" % m["train_words"] in r.stdout


def test_text_writer_equals_printf_lf(tmp_path):
    """ref :571 prints every value with fprintf("%lf "); the product formats in integer arithmetic (320 M values
    at cfg2).  Same bytes as libc's printf on ties (k/128 -> half-to-even), denormals, -0, huge values, inf, nan."""
    import ctypes.util
    libc = C.CDLL(ctypes.util.find_library("c"))
    libc.snprintf.restype = C.c_int
    c = w2b.Corpus(CORPUS, 1)
    V, D = c.vocab_size, 3000
    rng = np.random.default_rng(0)
    x = rng.integers(0, 2**32, V * D, dtype=np.uint64).astype(np.uint32).view(np.float32).copy()   # any bit pattern
    x[:4000] = (rng.integers(-10**6, 10**6, 4000) / 128.0).astype(np.float32)                       # exact ties
    x[4000:8000] = rng.standard_normal(4000).astype(np.float32) * np.float32(0.3)
    x[8000:8016] = [0.0, -0.0, np.inf, -np.inf, np.nan, -np.nan, 1e-45, -1e-45, 2.0**31, -2.0**31, 2.0**31 - 128,
                    3.4e38, 0.0000005, 0.0000015, 1 / 3, -1 / 3]
    x[8016:12000] = (rng.integers(0, 2**24, 3984) * 2.0 ** rng.integers(-30, 8, 3984)).astype(np.float32)
    out = str(tmp_path / "t.vec")
    _lib.check(w2b.lib().w2b_save_vectors(out.encode(), c._h, x.ctypes.data_as(_lib.f32p), D, 0))
    lines = open(out, "rb").read().split(b"\n")
    assert lines[0] == b"%d %d" % (V, D)
    buf = C.create_string_buffer(128)
    k = 0
    for a in range(V):
        toks = lines[1 + a].split(b" ")
        assert toks[0] == c.words()[a].encode() and toks[-1] == b"" and len(toks) == D + 2
        for b in range(D):
            v = float(x[k])
            if np.isfinite(v) and abs(v) < 1e15:
                want = b"%f" % v                      # Python's %f == glibc's %lf for finite doubles
            else:
                libc.snprintf(buf, 128, b"%lf", C.c_double(v))
                want = buf.value
            assert toks[1 + b] == want, (k, v, toks[1 + b], want)
            k += 1
    c.close()


def test_integration_patch_applies_compiles_and_fails_loudly_without_gpu(tmp_path):
    """INTEGRATION.md's reference-side patch, applied to a scratch copy of the reference source and linked against
    libword2bits_hip.so (oracle/make_integration_build.py): it must apply to the mounted reference, compile, and --
    here, without a GPU -- stop with the library's W2B_ENOGPU text instead of training on some CPU path."""
    import subprocess
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("no /root/reference here")
    import word2bits_amd as w2b
    subprocess.check_call([sys.executable, os.path.join(ROOT, "oracle", "make_integration_build.py")], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "oracle", "_ref", "word2bits_hipseam")
    assert os.path.exists(exe)
    if w2b.lib().w2b_device_count() > 0:
        pytest.skip("a GPU is visible: covered by tests/test_gpu_integration.py")
    r = subprocess.run([exe, "-train", os.path.join(ROOT, "tests", "golden", "corpus_small.txt"), "-output", str(tmp_path / "o"),
                        "-min-count", "3"], capture_output=True, text=True)
    assert r.returncode == 1 and "Vocab size: 60" in r.stdout and "no HIP device visible" in r.stdout


def test_tuning_struct_layout_matches_header():
    src = open(os.path.join(ROOT, "include", "word2bits_hip.h")).read()
    body = re.search(r"typedef struct w2b_tuning \{(.*?)\} w2b_tuning;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"int32_t\s+(\w+)", body)
    assert names == [n for n, _ in _lib.Tuning._fields_]
    assert C.sizeof(_lib.Tuning) == 4 * (12 + 4 + 4)          # round 6: + exchange_rule, exchange_tau_u / _v, concurrent_workers


def test_replica_token_slice_covers_what_the_workers_read():
    """./word2bits -gpus N uploads only a replica's part of the token stream (ref :377,414: a worker reads from its shard
    start until its word count passes the quota, then to the end of that sentence).  The slice rule of
    word2bits_amd.replicas (same as word2bits_main.cpp) against the oracle's reader: no worker of the replica ever
    reads a token outside [lo, hi), whatever the sub-sampling does (the reader stops at the first "</s>" behind its
    quota at the latest when nothing is dropped; sub-sampling only shortens what a sentence consumes)."""
    from word2bits_amd.replicas import replica_token_slice
    rng = np.random.default_rng(11)
    for trial in range(50):
        n = int(rng.integers(200, 4000))
        ids = rng.integers(1, 50, n).astype(np.int32)
        ids[rng.random(n) < 0.03] = 0                    # sentence ends
        workers = int(rng.integers(2, 9))
        quota = n // workers
        starts = np.sort(rng.integers(0, n, workers)).astype(np.int64)
        first = int(rng.integers(0, workers - 1))
        mine = starts[first:first + 2]
        lo, hi, more = replica_token_slice(ids, mine, quota)
        assert lo == mine.min() and hi <= n and more == (hi < n)
        for s in mine:                                     # what the reference's loop consumes (ref :394-423), no sub-sampling
            wc, cur = 0, int(s)
            while True:
                kept = 0
                while cur < n:                             # one sentence: up to "</s>" or 1000 kept words
                    tok = ids[cur]
                    cur += 1
                    wc += 1
                    if tok == 0:
                        break
                    kept += 1
                    if kept >= 1000:
                        break
                if cur >= n or wc > quota:
                    break
            assert cur <= hi, (trial, s, cur, hi)


def test_cli_warns_about_more_workers_than_the_corpus_supports(tmp_path):
    """a worker re-computes alpha only after 10 000 of its own words (ref :379-393): `-threads 512` on a 5 K-token file is
    accepted (as the reference accepts it) but warned about on stderr -- before any device is touched, so this runs here"""
    exe = os.path.join(ROOT, "word2bits")
    if not os.path.exists(exe):
        pytest.skip("CLI not built")
    out = str(tmp_path / "o.vec")
    r = subprocess.run([exe, "-train", CORPUS, "-output", out, "-threads", "512", "-min-count", "1"], capture_output=True, text=True)
    assert "warning: -threads 512" in r.stderr and "picks at most" in r.stderr
    r = subprocess.run([exe, "-train", CORPUS, "-output", out, "-threads", "12", "-min-count", "1"], capture_output=True, text=True)
    assert "warning" not in r.stderr


def test_cli_rejects_packed_output_of_unpackable_bitlevels_before_training(tmp_path):
    exe = os.path.join(ROOT, "word2bits")
    if not os.path.exists(exe):
        pytest.skip("CLI not built")
    out = str(tmp_path / "o.vec")
    for bl in ("0", "4"):
        r = subprocess.run([exe, "-train", CORPUS, "-output", out, "-packed", out + ".w2bp", "-bitlevel", bl], capture_output=True, text=True)
        assert r.returncode == 2 and "-packed needs -bitlevel 1 or 2" in r.stderr and "Starting epoch" not in r.stdout


def test_bench_names_the_workload_it_runs():
    """round 2 labelled every shape 'BASELINE configs[1]'; the label is built from the arguments now"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import argparse
    base = dict(vocab=400_000, dim=800, window=8, negative=24, bitlevel=1, ids="zipf", tokens=100_000_000)
    name = lambda **kw: bench.workload_name(argparse.Namespace(**dict(base, **kw)))
    assert name() == "BASELINE configs[1]"
    assert name(tokens=30_000_000).startswith("BASELINE configs[1] shape")
    assert "configs[4]" in name(vocab=3_700_000, dim=1000, negative=12)
    assert "configs[4]" in name(vocab=3_700_000, dim=1000, negative=12, bitlevel=0)
    assert "configs[0]" in name(vocab=60238, dim=200)
    assert "configs[2]" in name(vocab=60238, dim=400, bitlevel=2)
    assert name(ids="uniform") == "custom shape" and name(dim=640) == "custom shape"
    assert bench.algorithmic_bytes_per_word(800, 9, 24) == 217_736


def test_no_parked_patches():
    """round 3 kept an unmerged 414-line kernel patch under tools/patches/ with a test that it "keeps applying"; the review
    called it dead weight (land it or drop it).  It was dropped in round 4 (DESIGN.md section 8 says what it measured);
    experiments live on branches, not as patch files in the tree."""
    import glob
    assert not glob.glob(os.path.join(ROOT, "tools", "patches", "*.patch"))


# ------------------------------------------------------------------ round 4: the row rules, without a GPU (w2b_plan_rows)
def _plan(cn, workers, num_cus=256, D=800, window=8, negative=24, sample=0.0, **tune):
    import ctypes as C
    from word2bits_amd import _lib
    L = _lib.lib()
    cfg = _lib.Config()
    cfg.vocab_size, cfg.train_words, cfg.iter = len(cn), int(cn.sum()), 1
    cfg.layer1_size, cfg.window, cfg.negative, cfg.bitlevel, cfg.num_threads = D, window, negative, 1, workers
    cfg.alpha, cfg.sample = 0.05, sample
    tn = None
    if tune:
        tn = _lib.Tuning()
        tn.struct_size = C.sizeof(_lib.Tuning)
        tn.hot_rows_v = tn.hot_rows_u = tn.mem_mode = tn.atomic_rank = -1
        tn.hot_cap, tn.hot_weight_permille, tn.window_refresh = 128, 125, 16
        for k, v in tune.items():
            setattr(tn, k, v)
    out = _lib.RowPlan()
    cn = np.ascontiguousarray(cn, np.int64)
    _lib.check(L.w2b_plan_rows(C.byref(cfg), C.byref(tn) if tn is not None else None, cn.ctypes.data_as(_lib.i64p), num_cus, workers, C.byref(out)))
    return {k: getattr(out, k) for k, _ in _lib.RowPlan._fields_}


def test_row_rules_copies_only_on_a_full_device_lossless_context_rows_below():
    """DESIGN.md section 3.3a as arithmetic on the word counts (no GPU): a Zipf(1) vocabulary of 400 K words without sub-sampling
    (the benchmarked regime).  Below 3 workgroups per CU: no per-XCD copies, the context rows that at least a quarter of
    another worker holds at any moment are updated by atomic adds (a prefix that grows with the number of workers), target
    rows are plain.  On a full device: copies of both tables by the load rule, no adds."""
    V = 400_000
    cn = np.maximum((7.4e6 / np.arange(1, V + 1)).astype(np.int64), 5)      # ~100 M tokens, rank-1 word 7.4 %
    cn[0] = 100_000                                                          # "</s>"
    share = cn[1:] / cn[1:].sum()                                            # ("</s>" is never a context word, ref :400)
    prev = 0
    for workers in (8, 64, 256, 512, 767):
        p = _plan(cn, workers)
        # (round 5: between the reference's scale and 2.5 workgroups per CU -- 257 .. 640 workers -- four target rows get copies
        # that are merged every word; measured -0.05 % instead of +1.56 % at 512 workers)
        mid = 256 < workers <= 640
        assert p["full_device"] == 0 and p["copies_u"] == 0 and p["copies_v"] == (4 if mid else 0) and p["atomic_rank_v"] == 0
        assert p["merge_period"] == 1
        want = int(np.sum(workers * 9 * share >= 0.25))                    # workers x (window + 1) x share of the tokens >= 1/4
        assert abs(p["atomic_rank_u"] - want) <= 1 and p["atomic_rank_u"] >= prev
        prev = p["atomic_rank_u"]
    assert _plan(cn, 8)["atomic_rank_u"] == int(np.sum(8 * 9 * share >= 0.25)) > 0
    for workers in (768, 1024):
        p = _plan(cn, workers)
        assert p["full_device"] == 1 and p["copies_u"] > 50 and p["copies_v"] > 50 and p["merge_period"] == 16
        assert p["atomic_rank_u"] == 0 and p["atomic_rank_v"] == 0
    # explicit numbers win on either side of the threshold
    assert _plan(cn, 64, hot_rows_v=5, hot_rows_u=0)["copies_v"] == 5
    p = _plan(cn, 1024, hot_rows_v=0, hot_rows_u=0)
    assert p["copies_u"] == p["copies_v"] == 0 and p["atomic_rank_u"] > 1000          # shared rows on a full device: the adds are back
    assert _plan(cn, 256, atomic_rank_u=-1)["atomic_rank_u"] == 0
    # a smaller GPU: "full" is relative to its compute units
    assert _plan(cn, 256, num_cus=64)["full_device"] == 1


def test_row_rules_sub_sampling_and_flat_vocabularies():
    """With the reference's default sub-sampling the load of a context row is its share of the KEPT tokens (ref :403-406): the
    frequent words are thinned, the kept stream is shorter, and the prefix of rows that reach the quarter-of-a-worker load is a
    different (here: longer and flatter) one than without; a small flat vocabulary (the planted corpus: every row hit by
    several workers per window) updates BOTH tables by adds (round 3's rule, kept)."""
    V = 70_000
    cn = np.maximum((1.45e6 / np.arange(1, V + 1)).astype(np.int64), 5)
    cn[0] = 17_000
    a = _plan(cn, 256, D=200, sample=0.0)["atomic_rank_u"]
    b = _plan(cn, 256, D=200, sample=1e-3)["atomic_rank_u"]
    assert 0 < a < V - 1 and 0 < b < V - 1 and a != b
    st = 1e-3 * cn.sum()
    kept = np.minimum(cn[1:], np.sqrt(cn[1:] * st) + st)                          # expected kept occurrences (ref :403-406)
    assert abs(b - int(np.sum(256 * 9 * kept / kept.sum() >= 0.25))) <= 2
    flat = np.full(2129, 300, np.int64)
    p = _plan(flat, 64, D=200)
    assert p["atomic_rank_v"] == p["atomic_rank_u"] == 2128 and p["copies_v"] == 0
    assert _plan(flat, 2, D=200)["atomic_rank_v"] == 0


def test_row_rules_round5_kernel_choice_refreshed_copies_and_unsupported_shapes():
    """round 5, still pure host arithmetic: (a) the row-group kernel is the automatic choice for rows of at most 512 floats
    where the fidelity budget is not already thin -- not for shards below 50 000 words per worker, not for vocabularies so
    small and flat that every row collides, not at -size 800; (b) with it, only the very hottest context rows are read at
    refreshed copies (load rule 40: a handful at 256 workers, none at 32, never more than the rows with lossless adds);
    (c) ADVICE r04: shapes without an atomics-capable kernel (16-byte columns beyond 1024 floats, relaxed rows) report NO
    lossless rows, explicit ranks included -- the plan says what runs."""
    V = 70_000
    cn = np.maximum((1.45e6 / np.arange(1, V + 1)).astype(np.int64), 5)       # ~17 M tokens
    cn[0] = 17_000
    p = _plan(cn, 256, D=200)
    assert p["row_group_kernel"] == 1 and 2 <= p["refresh_rows_u"] <= 8 and p["refresh_rows_u"] <= p["atomic_rank_u"]
    assert _plan(cn, 32, D=200)["refresh_rows_u"] == 0 and _plan(cn, 32, D=200)["row_group_kernel"] == 1
    assert _plan(cn, 256, D=200, refresh_rows_u=-1)["refresh_rows_u"] == 0
    assert _plan(cn, 256, D=200, refresh_rows_u=64)["refresh_rows_u"] == 64
    assert _plan(cn, 256, D=800)["row_group_kernel"] == 0                      # long rows: the plain kernel
    assert _plan(cn, 512, D=200)["row_group_kernel"] == 0                      # 34 000 words per worker: shards too short (and copies)
    assert _plan(cn, 256, D=200, negative=30)["row_group_kernel"] == 0         # more targets than the groups hold
    flat = np.full(2129, 30_000, np.int64)
    assert _plan(flat, 8, D=400)["row_group_kernel"] == 0                      # every row collides: the budget is thin
    # (c)
    flat2 = np.full(2129, 300, np.int64)
    assert _plan(flat2, 64, D=800)["atomic_rank_v"] == 2128
    # (round 6) ... and on such a vocabulary the plain kernel runs 3/8 of the workers at a time, at least 16 (w2b_tuning.concurrent_workers)
    assert _plan(flat2, 64, D=400)["concurrent_workers"] == 24 and _plan(flat2, 512, D=200)["concurrent_workers"] == 192
    assert _plan(flat2, 16, D=400)["concurrent_workers"] == 16 and _plan(flat2, 8, D=400)["concurrent_workers"] == 8
    assert _plan(flat2, 64, D=400, concurrent_workers=40)["concurrent_workers"] == 40
    assert _plan(cn, 256, D=800)["concurrent_workers"] == 256 and _plan(cn, 1024, D=800)["concurrent_workers"] == 1024
    assert _plan(flat2, 64, D=1024, atomic_rank=100)["atomic_rank_v"] == 100     # the widest row with an ATOM instantiation
    p = _plan(flat2, 64, D=2048, atomic_rank=100, atomic_rank_u=100)
    assert p["atomic_rank_v"] == 0 and p["atomic_rank_u"] == 0
    assert _plan(flat2, 64, D=200, mem_mode=1, atomic_rank_u=100)["atomic_rank_u"] == 0
    assert _plan(flat2, 64, D=200, atomic_rank_u=100)["atomic_rank_u"] == 100


def test_suggested_exchange_interval():
    """./word2bits -gpus N: centre words per replica between two exchanges -- 1 / 32 of a replica's epoch, clamped to 32 K ... 1 M
    (w2b_suggested_exchange_words; DESIGN.md section 3.5 has the four regimes it was read off)"""
    f = w2b.lib().w2b_suggested_exchange_words
    assert f(22_021_995, 8) == 22_021_995 // 8 // 32 == 86_023          # the 22 M-token proxy: 86 K words
    assert f(100_099_995, 8) == 100_099_995 // 8 // 32 == 391_015           # configs[1] literally
    assert f(1_000_000_000, 8) == 1_048_576                              # configs[3]: the 1 M words the links allow
    assert f(1_000_000, 8) == 32_768 and f(100, 0) == 32_768
    assert f(100_099_995, 1) == 1_048_576
