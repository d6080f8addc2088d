"""-m gpu: the MI355X analogy evaluator (include/word2bits_eval.h) through the C ABI against the evaluator oracle
and against the committed stdout of both builds of the unmodified reference (tests/golden/eval_golden.json).
Integer/index work and float scores alike are compared BIT-EXACTLY: the kernel keeps the reference's summation
order and rounding (fused or not), so even massive ties resolve to the reference's answer."""
import json
import os
import subprocess

import numpy as np
import pytest

import word2bits_amd as w2b
from w2b_testlib import GOLDEN, ROOT, eval_oracle, write_vectors_file

pytestmark = pytest.mark.gpu
ALL = json.load(open(os.path.join(GOLDEN, "eval_golden.json")))
RUNS = [g for g in ALL if "vectors" in g]
CLI = os.path.join(ROOT, "compute_accuracy")


def same_floats(a, b):
    """bit-identical except that any NaN matches any NaN (x86 0/0 is -nan, the GPU's is +nan)"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    nan = np.isnan(a) & np.isnan(b)
    return np.array_equal(a.view(np.uint32)[~nan], b.view(np.uint32)[~nan]) and np.array_equal(np.isnan(a), np.isnan(b))


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("vec,bitlevel,threshold", [("eval_1bit.bin", 0, 0), ("eval_1bit.bin", 0, 100),
                                                    ("eval_fp.bin", 0, 0), ("eval_fp.bin", 1, 0),
                                                    ("eval_fp.bin", 2, 0), ("eval_fp.bin", 4, 0),
                                                    ("eval_fp.bin", 3, 150), ("eval_fp.bin", 8, 0)])
def test_load_normalise_and_top1_bit_exact_on_fixtures(gpu, vec, bitlevel, threshold, fused):
    E = eval_oracle()
    path = os.path.join(GOLDEN, vec)
    om = E.EvalModel(path, bitlevel, threshold, fma=fused)
    ev = w2b.Evaluator(path, bitlevel, threshold, fused=fused)
    assert (ev.words, ev.size) == (om.words, om.size)
    assert [ev.word(i) for i in range(ev.words)] == om.names
    for w in (b"THE", b"X" * 50, b"Y" * 50, b"NOPE", b"</S>"):
        assert ev.lookup(w) == om.lookup(w)
    assert same_floats(ev.matrix(), om.M)
    rng = np.random.default_rng(5)
    b = rng.integers(0, ev.words, (3, 400)).astype(np.int32)
    b[:, :20] = b[0, :20]                      # b1 == b2 == b3
    got, gd = ev.top1(*b)
    want, wd = om.top1(*b)
    assert np.array_equal(got, want)
    assert same_floats(gd, wd)
    ev.close()


@pytest.mark.parametrize("build", ["compute_accuracy", "compute_accuracy_nofma"])
def test_transcripts_equal_reference_stdout(gpu, build):
    evs = {}
    for g in RUNS:
        if g["build"] != build:
            continue
        key = (g["vectors"], g["bitlevel"], g["threshold"])
        if key not in evs:
            evs[key] = w2b.Evaluator(os.path.join(GOLDEN, g["vectors"]), g["bitlevel"], g["threshold"],
                                     fused=(build == "compute_accuracy"))
        got = evs[key].transcript(open(os.path.join(GOLDEN, g["questions"]), "rb").read())
        assert got.decode("latin1") == g["stdout"], key + (g["questions"],)
    for e in evs.values():
        e.close()


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("kind,V,D,Q", [("1bit", 3000, 200, 700), ("2bit", 1500, 400, 300), ("fp", 5000, 300, 300),
                                        ("fp", 777, 1000, 130), ("1bit", 129, 5, 257)])
def test_top1_bit_exact_on_seeded_inputs(gpu, kind, V, D, Q, fused, tmp_path):
    """sizes that cross tile edges (rows % 128, questions % 128, size % 16 all != 0) and the question-row
    exclusions; 1-bit / 2-bit inputs tie on most questions"""
    E = eval_oracle()
    rng = np.random.default_rng(V + D)
    if kind == "1bit":
        M = (rng.integers(0, 2, (V, D)) * 2 - 1).astype(np.float32) / np.float32(3)
    elif kind == "2bit":
        M = (rng.choice([.25, .75], (V, D)) * rng.choice([-1, 1], (V, D))).astype(np.float32)
    else:
        M = (rng.standard_normal((V, D)) * rng.choice([1e-3, 1, 30], (V, 1))).astype(np.float32)
    names = [("w%d" % i).encode() for i in range(V)]
    path = write_vectors_file(str(tmp_path / "v.bin"), names, M)
    om, ev = E.EvalModel(path, 0, 0, fma=fused), w2b.Evaluator(path, 0, 0, fused=fused)
    assert same_floats(ev.matrix(), om.M)
    b = rng.integers(0, V, (3, Q)).astype(np.int32)
    got, gd = ev.top1(*b)
    want, wd = om.top1(*b)
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
    assert same_floats(gd, wd)
    if fused:              # the same fused chain on the vector ALU (w2b_eval_set_kernel 0): a cross-check of the MFMA path
        ev.set_kernel(0)
        got0, gd0 = ev.top1(*b)
        assert np.array_equal(got0, want) and same_floats(gd0, wd)
        ev.set_kernel(1)
    if kind != "fp":
        # the fixture really is tie-dominated: the best score is shared by several rows for most questions
        q = 0
        vec = (om.M[b[1, q]] - om.M[b[0, q]]) + om.M[b[2, q]]
        assert (np.abs(om.M @ vec - wd[q]) < 1e-6).sum() >= 1
    ev.close()


def test_full_size_properties(gpu, tmp_path):
    """text8-sized vocabulary (60238 x 200, 1-bit), 70000 questions (two launches): answers that are known
    without a reference.  For b1 == b2 the query is exactly row b3, so the best row is the lowest-numbered
    duplicate of row b3 when one is planted (score |row|^2, strictly above every non-identical row), and
    running the same questions twice or in a different order gives the same answers."""
    V, D, Q = 60238, 200, 70000
    rng = np.random.default_rng(11)
    M = (rng.integers(0, 2, (V, D)) * 2 - 1).astype(np.float32) / np.float32(3)
    dup_src = rng.choice(np.arange(1000, V), 500, replace=False)
    dup_dst = np.arange(200, 700)
    M[dup_dst] = M[dup_src]                       # rows 200..699 are copies of later rows
    names = [("w%d" % i).encode() for i in range(V)]
    path = write_vectors_file(str(tmp_path / "v.bin"), names, M)
    ev = w2b.Evaluator(path, 0, 0)
    x = rng.integers(0, V, Q).astype(np.int32)
    b3 = rng.choice(dup_src, Q).astype(np.int32)
    best, bestd = ev.top1(x, x, b3)
    where = {int(s): int(d) for s, d in zip(dup_src, dup_dst)}
    want = np.array([where[int(s)] for s in b3], np.int32)
    ok = x != want                                # the duplicate itself may be an excluded question row
    assert np.array_equal(best[ok], want[ok])
    assert np.all(bestd[ok] > 0.99) and np.all(bestd[ok] < 1.01)
    perm = rng.permutation(Q)
    best2, bestd2 = ev.top1(x[perm], x[perm], b3[perm])
    assert np.array_equal(best2, best[perm]) and np.array_equal(bestd2.view(np.uint32), bestd[perm].view(np.uint32))
    ms, launches, macs = ev.timing()
    assert launches == 4 and ms > 0 and macs >= 2.0 * Q * V * D
    ev.close()


def test_cli_stdout_equals_reference(gpu):
    for g in RUNS:
        if g["questions"] == "eval_q_empty.txt" or g["bitlevel"] not in (0, 2):
            continue
        args = [CLI, os.path.join(GOLDEN, g["vectors"]), str(g["bitlevel"]), str(g["threshold"])]
        if g["build"] == "compute_accuracy_nofma":
            args.append("nofma")
        r = subprocess.run(args, stdin=open(os.path.join(GOLDEN, g["questions"]), "rb"), capture_output=True)
        assert r.returncode == 0, r.stderr[-300:]
        assert r.stdout.decode("latin1") == g["stdout"], args


@pytest.mark.parametrize("bitlevel", [1, 2])
def test_train_then_evaluate_end_to_end(gpu, bitlevel, tmp_path):
    """./word2bits (GPU) -> vectors file -> ./compute_accuracy (GPU): the evaluator's stdout on really trained,
    quantized vectors equals the CPU restatement's (and the unmodified reference evaluator's where its binary
    travelled), and the planted analogies are actually found."""
    from planted import make_planted, parse_accuracy
    from w2b_testlib import ref_binary
    corpus, questions, out = (str(tmp_path / n) for n in ("c.txt", "q.txt", "v.bin"))
    make_planted(corpus, questions, sections=8, pairs=12, repeats=60, seed=3)
    r = subprocess.run([os.path.join(ROOT, "word2bits"), "-train", corpus, "-output", out, "-bitlevel", str(bitlevel),
                        "-size", "200", "-window", "8", "-negative", "24", "-threads", "64", "-iter", "5",
                        "-min-count", "5", "-binary", "1", "-eval", questions], capture_output=True)
    assert r.returncode == 0, r.stderr[-300:]
    qs = open(questions, "rb").read()
    got = subprocess.run([CLI, out, "0", "0"], input=qs, capture_output=True)
    assert got.returncode == 0, got.stderr[-300:]
    assert r.stdout.endswith(got.stdout)           # ./word2bits -eval prints the same transcript after training
    E = eval_oracle()
    assert got.stdout == E.transcript(E.EvalModel(out, 0, 0, fma=True), qs)
    exe = ref_binary("compute_accuracy")
    if exe:
        assert got.stdout == subprocess.run([exe, out, "0", "0"], input=qs, capture_output=True).stdout
    acc = parse_accuracy(got.stdout.decode())
    assert acc["seen"] == acc["questions"] == 8 * 12 * 11 and acc["total"] > 5.0, acc


def _fuzz_model(tmp_path_factory):
    rng = np.random.default_rng(77)
    V, D = 40, 12
    names = [b"</s>"] + [n.encode() for n in ("aa ab ac ad ba bb bc bd ca cb cc cd da db dc dd The the THE exit "
                                              "x1 x2 x3 x4 x5 x6 x7 x8 x9 y1 y2 y3 y4 y5 y6 y7 y8 y9 zz").split()]
    names = names[:V]
    M = (rng.integers(0, 2, (V, D)) * 2 - 1).astype(np.float32) / np.float32(3)
    return write_vectors_file(str(tmp_path_factory.mktemp("fz") / "v.bin"), names, M), names


def test_question_stream_state_machine_fuzz(gpu, tmp_path_factory):
    """random token soups through the scanf-style loop (ref :113-188): section markers anywhere, EXIT, unknown
    words, questions cut short by the end of the input, missing trailing white space -- same stdout as the oracle"""
    from hypothesis import given, settings, strategies as st
    E = eval_oracle()
    path, names = _fuzz_model(tmp_path_factory)
    om, ev = E.EvalModel(path, 0, 0, fma=True), w2b.Evaluator(path, 0, 0, fused=True)
    words = [n.decode() for n in names[1:24]] + [":", ":", "EXIT", "exit", "nope", "Aa", "BB"]
    seps = [" ", " ", "\n", "\n", "\t", "  ", "\r\n", " \n "]

    @settings(max_examples=150, deadline=None, derandomize=True, database=None)
    @given(st.lists(st.tuples(st.sampled_from(words), st.sampled_from(seps)), min_size=0, max_size=60),
           st.booleans())
    def run(tokens, trailing):
        text = "".join(w + s for w, s in tokens)
        if not trailing:
            text = text.rstrip()
        q = text.encode()
        assert ev.transcript(q) == E.transcript(om, q), text

    run()
    ev.close()


def test_vector_file_reader_fuzz(gpu, tmp_path_factory):
    """random vector files through the reader of ref :85-112: names with embedded newlines / tabs / upper case /
    51+ characters / bytes >= 0x80, duplicate names, truncated last row, header with extra white space, threshold"""
    from hypothesis import given, settings, strategies as st
    E = eval_oracle()
    d = tmp_path_factory.mktemp("vf")
    name_st = st.lists(st.sampled_from(list("abAB\n\t_") + ["\xe9", "x" * 26]), min_size=1, max_size=4).map("".join)

    @settings(max_examples=60, deadline=None, derandomize=True, database=None)
    @given(st.lists(name_st, min_size=1, max_size=9), st.integers(1, 5), st.integers(0, 12), st.integers(0, 40),
           st.sampled_from(["%d %d\n", "%d  %d\n", " %d\n%d\n"]), st.integers(0, 3))
    def run(names, D, threshold, cut, header, seed):
        rng = np.random.default_rng(seed)
        V = len(names)
        M = rng.standard_normal((V, D)).astype(np.float32)
        p = str(d / "v.bin")
        with open(p, "wb") as f:
            f.write((header % (V, D)).encode())
            for n, row in zip(names, M):
                f.write(n.encode("latin1") + b" " + row.tobytes() + b"\n")
        if cut:
            data = open(p, "rb").read()
            open(p, "wb").write(data[:max(len(header), len(data) - cut)])
        om, ev = E.EvalModel(p, 0, threshold, fma=True), w2b.Evaluator(p, 0, threshold, fused=True)
        try:
            assert (ev.words, ev.size) == (om.words, om.size)
            assert [ev.word(i) for i in range(ev.words)] == om.names
            assert same_floats(ev.matrix(), om.M)
            if ev.words >= 1:
                b = rng.integers(0, ev.words, (3, 8)).astype(np.int32)
                got, gd = ev.top1(*b)
                want, wd = om.top1(*b)
                assert np.array_equal(got, want) and same_floats(gd, wd)
        finally:
            ev.close()

    run()


@pytest.mark.parametrize("bitlevel,ev_bitlevel,threshold,fused", [(1, 0, 0, True), (2, 0, 0, False), (0, 2, 37, True)])
def test_evaluator_on_live_trainer_equals_file_round_trip(gpu, bitlevel, ev_bitlevel, threshold, fused, tmp_path):
    """w2b_eval_from_trainer (no file): names through the same reader logic (a 57-character word, a word with bytes >=
    0x80), quantize(u+v) exported on the device, the evaluator's own quantize / threshold / normalisation -- everything
    bit-identical to saving the vectors and loading the file."""
    rng = np.random.default_rng(bitlevel * 7 + threshold)
    V, D = 90, 44
    words = ["</s>"] + ["w%d" % i for i in range(1, V)]
    words[5] = "x" * 57
    words[6] = "caf\xe9"
    words[40] = "Mixed_Case"
    corpus = str(tmp_path / "c.txt")
    toks = rng.integers(1, V, 6000)
    with open(corpus, "wb") as f:
        for i in range(0, len(toks), 20):
            f.write(" ".join(words[t] for t in toks[i:i + 20]).encode("latin1") + b"\n")
    c = w2b.Corpus(corpus, 1)
    t = w2b.Trainer(c.vocab_size, D, 5, 5, bitlevel, num_threads=4, iter=1, sample=0.0, train_words=c.train_words)
    t.init_net()
    t.set_vocab_counts(c.counts(), 100000)
    t.set_corpus(c.tokens())
    starts, ov = c.shards(4)
    t.set_shards(starts, ov)
    t.train_epoch(500)
    path = str(tmp_path / "v.bin")
    c.save_vectors(path, t.export_quantized(), 1)
    a = w2b.Evaluator(path, ev_bitlevel, threshold, fused=fused)
    b = w2b.Evaluator.from_trainer(t, c.words(), ev_bitlevel, threshold, fused=fused)
    try:
        assert (a.words, a.size) == (b.words, b.size)
        assert [a.word(i) for i in range(a.words)] == [b.word(i) for i in range(b.words)]
        assert same_floats(a.matrix(), b.matrix())
        qs = (": s\n" + "".join("%s %s %s %s\n" % tuple(c.words()[j] for j in rng.integers(1, c.vocab_size, 4))
                                for _ in range(300))).encode("latin1")
        assert a.transcript(qs) == b.transcript(qs)
    finally:
        a.close(); b.close(); t.close(); c.close()
