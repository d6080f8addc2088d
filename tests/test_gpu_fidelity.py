"""-m gpu: Hogwild fidelity of the kernels users actually run (SURVEY 8c rung 6).

With several workers nothing is bit-reproducible -- not in the reference either -- so these tests hold the HIP trainer
to what the UNMODIFIED reference program does with the SAME number of Hogwild threads on the same corpus:
`tests/golden/fidelity_golden.json` (planted-analogy corpus; generator make_fidelity_golden.py) and
`tests/golden/fidelity_text8size.json` (17 M-token text8-sized corpus; make_fidelity_golden_text8size.py) record its
per-epoch "Epoch Loss" values and, for the planted corpus, the total accuracy printed by the unmodified evaluator, over
several runs per thread count.  Asserted here, per worker count:
  * every epoch loss of `./word2bits` (default kernel = sentence-resident, coherent rows) within LOSS_RTOL[workers] of
    the mean of the reference's runs with that many threads, and within RESIDENT_VS_PLAIN_RTOL of the plain worker kernel;
  * total accuracy (scored by ./compute_accuracy, whose transcript must equal the unmodified evaluator's byte for byte
    when oracle/_ref is present) inside the reference's own band widened by ACC_MARGIN points.
The thread count matters as much as the implementation (the reference's own last-epoch loss moves from -401 K to -395 K to
-383 K between 8, 64 and 512 threads on the planted corpus, its accuracy from 44 % to 49 % to 77 %: each thread
re-computes alpha only after 10 000 of its own words), so the comparison is always at equal counts.
BASELINE configs[2] (text8, bitlevel 2, size 400, negative 24, iter 5 + compute-accuracy parity) is covered at its own
shape on the planted corpus -- text8 and questions-words.txt are not available offline."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from w2b_testlib import GOLDEN, ROOT, ref_binary
from planted import make_planted, parse_accuracy

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "word2bits")
EVAL = os.path.join(ROOT, "compute_accuracy")
GOLD = json.load(open(os.path.join(GOLDEN, "fidelity_golden.json")))

# Tolerances, calibrated on MI355X against the committed reference runs (values of round 2 in DESIGN.md section 6):
#  * up to 8 workers both kernels track the reference's epoch losses within 1 %;
#  * at 64 workers on this 3 310-word, 564 K-token corpus (8 800 words per worker and epoch) a GPU really runs 64 workers at
#    once where the 8-core host that produced the bands time-slices 64 threads: the plain kernel's first epoch is 5 %
#    off and the later ones < 2 %; the sentence-resident kernel (rows stay on chip for up to 17 positions) 8 % / < 4.5 %;
#    at the cfg2 shape (bitlevel 2, size 400) both kernels end 3-6 % BETTER than the reference's 64-thread losses.
#  * total accuracy on this corpus moves by several points from run to run on either side (reference: 42.0-47.0 % over
#    1 / 8 / 64 threads; HIP at 8 workers: 41.3-50.3 % over the runs of this round), hence the wide margin.
LOSS_RTOL = {8: (0.02, 0.02), 64: (0.09, 0.08)}        # workers -> (first epoch, later epochs), vs the reference mean
RESIDENT_VS_PLAIN_RTOL = 0.045                         # sentence-resident vs plain worker kernel, same worker count
ACC_MARGIN = 8.0                                       # points of total accuracy around the reference's [min, max] band


@pytest.fixture(scope="module")
def planted(tmp_path_factory):
    d = tmp_path_factory.mktemp("planted")
    corpus, questions = str(d / "planted.txt"), str(d / "questions.txt")
    ntok = make_planted(corpus, questions, repeats=120)
    assert ntok == GOLD["corpus"]["tokens"]
    return corpus, questions, d


def train(corpus, out, threads, flags, extra=()):
    args = [CLI, "-train", corpus, "-output", out, "-threads", str(threads), "-min-count", "5", "-binary", "1"]
    for k, v in flags.items():
        args += ["-" + k, str(v)]
    r = subprocess.run(args + list(extra), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-300:] + r.stderr[-300:]
    return [float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", r.stdout)]


def score(vec, questions):
    with open(questions, "rb") as q:
        qs = q.read()
    got = subprocess.run([EVAL, vec, "0", "0"], input=qs, capture_output=True).stdout
    ref = ref_binary("compute_accuracy")
    if ref:                                     # compute-accuracy parity: the product's transcript == the reference's
        want = subprocess.run([ref, vec, "0", "0"], input=qs, capture_output=True).stdout
        assert got == want
    return parse_accuracy(got.decode())


def reference_band(config, threads):
    runs = [r for r in GOLD["configs"][config]["runs"] if r["threads"] == threads]
    assert runs
    losses = np.array([r["epoch_losses"] for r in runs])
    acc = [r["accuracy"]["total"] for r in runs]
    return losses.mean(axis=0), min(acc), max(acc)


def check_against_reference(config, threads, planted, label=""):
    corpus, questions, d = planted
    flags = GOLD["configs"][config]["flags"]
    want, acc_lo, acc_hi = reference_band(config, threads)
    res = {}
    for name, extra in (("resident", ["-window-cache", "1"]), ("plain", ["-window-cache", "0"])):
        out = str(d / ("%s_%s_%d.bin" % (config, name, threads)))
        losses = train(corpus, out, threads, flags, extra)
        acc = score(out, questions)
        assert acc["seen"] == acc["questions"] == 7728
        res[name] = (np.array(losses), acc["total"])
        print("FIDELITY %s threads=%d %s: losses %s acc %.2f | reference mean %s acc band [%.2f, %.2f]" %
              (config, threads, name, np.round(losses).tolist(), acc["total"], np.round(want).tolist(), acc_lo, acc_hi))
    first, later = LOSS_RTOL[threads]
    tol = np.array([first] + [later] * (len(want) - 1))
    for name, (losses, acc) in res.items():
        assert len(losses) == len(want)
        assert np.all(np.abs(losses - want) <= tol * np.abs(want)), (name, losses.tolist(), want.tolist())
        assert acc_lo - ACC_MARGIN <= acc <= acc_hi + ACC_MARGIN, (name, acc, acc_lo, acc_hi)
    r, p = res["resident"][0], res["plain"][0]
    assert np.all(np.abs(r - p) <= RESIDENT_VS_PLAIN_RTOL * np.abs(p)), (r.tolist(), p.tolist())


@pytest.mark.parametrize("threads", [8, 64])
def test_planted_1bit_matches_reference_at_equal_thread_count(gpu, planted, threads):
    """bitlevel 1, size 200, window 8, negative 24, iter 5 (BASELINE configs[0] shape, 5 epochs)"""
    check_against_reference("b1_d200", threads, planted)


@pytest.mark.parametrize("threads", [8, 64])
def test_cfg2_shape_2bit_d400_accuracy_parity(gpu, planted, threads):
    """BASELINE configs[2] shape: bitlevel 2, size 400, negative 24, iter 5 -- epoch losses and compute-accuracy
    parity against the reference CPU program at the same thread count"""
    check_against_reference("cfg2_b2_d400", threads, planted)


def test_planted_512_workers_bounded(gpu, planted):
    """512 workers on a 564 K-token corpus is a regime `-threads 0` never selects (1 100 words per worker: alpha is
    never re-computed, ref :379-393) -- the reference's own 512-thread runs train at the starting alpha for all five
    epochs (last-epoch loss -383 K against -402 K with 8 threads, accuracy 77 % against 44 %).  A GPU runs the 512
    workers truly concurrently on 3 310 rows where the 8-core reference host time-slices them, so only the end state
    is bounded: last-epoch loss within 15 % of the reference's 512-thread mean, every epoch better than the one before."""
    corpus, questions, d = planted
    flags = GOLD["configs"]["b1_d200"]["flags"]
    want, acc_lo, acc_hi = reference_band("b1_d200", 512)
    for name, extra in (("resident", ["-window-cache", "1"]), ("plain", ["-window-cache", "0"])):
        out = str(d / ("w512_%s.bin" % name))
        losses = np.array(train(corpus, out, 512, flags, extra))
        acc = score(out, questions)["total"]
        print("FIDELITY b1_d200 threads=512 %s: losses %s acc %.2f | reference mean %s acc band [%.2f, %.2f]" %
              (name, np.round(losses).tolist(), acc, np.round(want).tolist(), acc_lo, acc_hi))
        assert abs(losses[-1] - want[-1]) <= 0.15 * abs(want[-1]), (name, losses.tolist(), want.tolist())
        assert np.all(np.diff(losses) > 0), (name, losses.tolist())


def test_text8_size_threads0_resident_vs_plain_vs_reference(gpu, tmp_path_factory):
    """17 M tokens, 70 K words, bitlevel 1, size 200, window 8, negative 24, 3 epochs with `-threads 0` (as many workers as
    the GPU holds, capped at train_words / 20000): the default (sentence-resident) kernel against the plain coherent
    kernel and against the unmodified reference program's epoch losses on the same file."""
    from w2b_testlib import write_zipf_text_corpus
    gold = json.load(open(os.path.join(GOLDEN, "fidelity_text8size.json")))
    d = tmp_path_factory.mktemp("t8")
    corpus = write_zipf_text_corpus(str(d / "c.txt"))
    flags = dict(bitlevel=1, size=200, window=8, negative=24, iter=3)
    res = {}
    for name, extra in (("resident", ["-window-cache", "1"]), ("plain", ["-window-cache", "0"])):
        res[name] = np.array(train(corpus, str(d / "o.bin"), 0, flags, extra))
        print("FIDELITY text8size threads=0 %s: %s | reference (%d threads) %s" %
              (name, np.round(res[name]).tolist(), gold["threads"], np.round(gold["epoch_losses"]).tolist()))
    want = np.array(gold["epoch_losses"])
    # 850 concurrent workers against the reference's 8 threads.  Measured: plain kernel -57.5 / -54.3 / -53.4 M, sentence-
    # resident kernel with its (at most four) private hot target rows merged every 8 steps about -60 / -55.5 / -54 M,
    # reference -59.0 / -55.6 / -54.7 M.  (The window rows cost nothing here: with W2B_HOT_ROWS=0 the resident kernel
    # reproduces the plain kernel's losses; what moves the first epoch is the merge period of the private hot rows --
    # -64 M at 32 steps, -59 M at 4 -- see DESIGN.md section 6.)
    tol = np.array([0.05, 0.03, 0.03])
    for name, losses in res.items():
        assert np.all(np.abs(losses - want) <= tol * np.abs(want)), (name, losses.tolist(), want.tolist())
    assert np.all(np.abs(res["resident"] - res["plain"]) <= np.array([0.07, 0.035, 0.025]) * np.abs(res["plain"]))


def test_text8_size_window_residency_alone_is_loss_neutral(gpu, tmp_path_factory):
    """same corpus, 128 workers, private hot rows switched off: what remains of the sentence-resident kernel (LDS window,
    scratch entries, exact-or-merge write-back) must give the plain kernel's epoch losses (measured: -58.667 M vs
    -58.675 M in the first epoch)"""
    from w2b_testlib import write_zipf_text_corpus
    d = tmp_path_factory.mktemp("t8b")
    corpus = write_zipf_text_corpus(str(d / "c.txt"))
    flags = dict(bitlevel=1, size=200, window=8, negative=24, iter=1)
    r = np.array(train(corpus, str(d / "o.bin"), 128, flags, ["-window-cache", "1", "-hot-rows", "0"]))
    p = np.array(train(corpus, str(d / "o.bin"), 128, flags, ["-window-cache", "0", "-hot-rows", "0"]))
    print("FIDELITY text8size threads=128 hot rows off: resident %s plain %s" % (r.tolist(), p.tolist()))
    assert np.all(np.abs(r - p) <= 0.005 * np.abs(p))
