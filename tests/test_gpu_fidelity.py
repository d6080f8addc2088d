"""-m gpu: Hogwild fidelity of the kernels users actually run (SURVEY 8c rung 6).

With several workers nothing is bit-reproducible -- not in the reference either -- so these tests hold `./word2bits`
to what the UNMODIFIED reference program does with the SAME number of truly concurrent threads on the same corpus:
`tests/golden/fidelity_bands.json`, recorded by tests/golden/make_fidelity_bands.py on the GPU box's HOST (2 x EPYC
9575F, 256 hardware threads), 2-3 runs per thread count:
  planted_*        planted-analogy corpus (564 K tokens), bitlevel 1 size 200 and BASELINE configs[2] shape (bitlevel 2, size
                   400, negative 24, iter 5), 8 / 64 / 512 threads, scored by the unmodified evaluator
  text8size        17 M Zipf tokens over 70 K words, bitlevel 1 size 200 (BASELINE configs[0] shape), 3 epochs, 64 / 256 threads
  headline         the BENCHMARKED regime, BASELINE configs[1]: V = 400 K, size 800, window 8, negative 24, bitlevel 1,
                   -sample 0, 22 M tokens (every word 5x + a 20 M-token Zipf(1) stream), 64 / 256 threads
  heldout_k5       (round 4) HELD OUT -- no knob of the library was ever swept on it: V = 100 K, size 300, window 5, negative 5,
                   bitlevel 1, -sample 0, 8.5 M tokens, 64 / 256 threads
  heldout_zipf12   (round 4) HELD OUT: Zipf exponent 1.2 at the configs[2] shape (bitlevel 2, size 400, negative 24), default
                   -sample, 2 epochs, 64 / 256 threads

Every tolerance is  max(3 sigma of the reference's own runs at that thread count, FLOOR)  per epoch.  The reference is
extremely repeatable (sigma 0.01-0.4 % of an epoch loss), so the floor decides: ONE floor for every regime since round 4,
1.5 % (round 3 had 1.5 % / 3.5 % per regime, set just above what the product measured); rounds 3-5 stated one exception (the
planted corpus at the configs[2] shape with 64 workers), round 6 retired it (FLOOR_EXCEPTION below is empty).  What the product does to stay
inside it (profiles/r04_sessions/ has every matrix these numbers were read from):
  * the automatic kernel is the plain one (the sentence-resident kernel keeps context rows private for up to 2 x window + 1
    positions: -13 % on heldout_zipf12 at 256 workers; it is an explicit choice now and held to its own, looser bound below);
  * below a full device no row has per-XCD copies -- every row is shared by all workers as in the reference -- and the
    context rows are updated by lossless atomic adds (the reference's `u[c] += e[c]`, ref :500-502): measured
    +0.2 / +0.8 % at 64 / 256 workers in the benchmarked regime (440 workers: +1.3 ... +1.5 %) (round 3's rules: -2.2 / -2.8 / -1.4 %), within
    0.8 % everywhere on the two held-out regimes and the text8-sized corpus;
  * on a full device (>= 3 workgroups per CU: `-threads 0` on a corpus of 50 M words and more) the hottest rows would queue
    at their memory lines (13 M words/s instead of 28 M), so they get per-XCD copies kept together by consensus merges:
    -0.1 ... +0.4 % in the benchmarked regime at 1024 workers -- a measured balance of two opposite errors (updates lost
    inside an XCD, stale copies between XCDs), not a derived property; asserted on that regime and, once, on a held-out
    regime at that scale (test_full_device_on_a_held_out_regime: +0.6 %);
  * `-threads 0` never picks fewer than 50 000 words per worker and epoch (20 000 until round 3: the text8-sized corpus then
    ran 850 workers and its later epochs ended 2 % off whatever the row-update scheme), and when that is not a full device
    it stays at 256 workers, the reference's own scale (440 workers on the benchmarked regime's file: +1.3 ... +1.5 %).
Accuracy is asserted inside the reference's band widened by max(2 points, 3 sigma) (round 3: 5 points).
The product's `./compute_accuracy` transcript must equal the unmodified evaluator's byte for byte on every trained file.
text8 and questions-words.txt are not available offline; the planted corpus stands in for them."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from w2b_testlib import GOLDEN, ROOT, ref_binary, HELDOUT
from planted import make_planted, parse_accuracy

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "word2bits")
EVAL = os.path.join(ROOT, "compute_accuracy")
BANDS = json.load(open(os.path.join(GOLDEN, "fidelity_bands.json")))["jobs"]

FLOOR = 0.015            # of the reference's mean epoch loss; every regime, every worker count, the automatic kernel
# Rounds 3-5 had ONE stated exception: the planted corpus at the configs[2] shape (2 bits, size 400) with 64 workers -- 2 129 words
# without a frequency skew, every row hit by several workers per window: -1.0 ... -2.8 % over the five epochs (3.5 % allowed) and an
# accuracy of 19.2-20.5 against the reference's 16.9-17.8 (5 points allowed).  A GPU workgroup has a chunk of 13 target rows open for
# ~10 us where the reference's thread has one row open for ~1.5 us, so 64 workers AT ONCE collide far more often than 64 threads.
# Round 6 retired it: on such vocabularies the plain kernel runs 3/8 of the workers at a time (w2b_tuning.concurrent_workers,
# automatic; every worker still walks its own shard with its own LCG stream): +0.5 ... -0.9 %, accuracy 15.3-16.0
# (profiles/r06_sessions/r06h_planted_concurrency.txt, r06i).  No exception is left for the automatic kernel.
FLOOR_EXCEPTION = {}
ACC_EXCEPTION = {}
# (the sentence-resident kernel -- explicit only, not sliced -- keeps round 5's bound on that one case: -1.8 ... -2.5 %)
FLOOR_RESIDENT_EXCEPTION = {("planted_cfg2_b2_d400", 64): 0.035}
# The sentence-resident kernel (explicit: -window-cache 1) where it was measured (profiles/r03_sessions, r04_sessions):
# planted corpus and text8-sized corpus within 2.5 %; NOT asserted on the held-out regimes (-13 % on heldout_zipf12).
FLOOR_RESIDENT = 0.025
ACC_POINTS = 2.0


def band(job, threads):
    runs = [r for r in BANDS[job]["runs"] if r["threads"] == threads]
    assert len(runs) >= 2
    L = np.array([r["epoch_losses"] for r in runs])
    acc = np.array([r["accuracy"]["total"] for r in runs]) if "accuracy" in runs[0] else None
    return L.mean(0), L.std(0, ddof=1), acc


def loss_tolerance(job, threads, floor=FLOOR):
    mean, std, _ = band(job, threads)
    return np.maximum(3 * std, floor * np.abs(mean))


def train(corpus, out, threads, flags, extra=()):
    # (W2B_FIDELITY_EXTRA: extra command-line flags for every run -- how the builder sessions re-run these tests under a
    # knob arm, e.g. "-atomic-rank 0"; never set by the driver)
    hook = os.environ.get("W2B_FIDELITY_EXTRA", "").split()
    if hook:                                            # never silent: a leaked variable would change what the gates measure
        print("FIDELITY WARNING: W2B_FIDELITY_EXTRA adds %s to this run -- a builder session's knob arm, NOT the shipped defaults" % hook)
    extra = list(extra) + hook
    args = [CLI, "-train", corpus, "-output", out, "-threads", str(threads), "-min-count", "5", "-binary", "1"]
    for k, v in flags.items():
        args += ["-" + k, str(v)]
    r = subprocess.run(args + list(extra), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-300:] + r.stderr[-300:]
    w = re.search(r"Hogwild workers \(workgroups\): (\d+)", r.stdout)
    return np.array([float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", r.stdout)]), (int(w.group(1)) if w else threads), r.stderr


def score(vec, questions):
    with open(questions, "rb") as q:
        qs = q.read()
    got = subprocess.run([EVAL, vec, "0", "0"], input=qs, capture_output=True).stdout
    ref = ref_binary("compute_accuracy")
    if ref:                                     # compute-accuracy parity: the product's transcript == the reference's
        want = subprocess.run([ref, vec, "0", "0"], input=qs, capture_output=True).stdout
        assert got == want
    return parse_accuracy(got.decode())


def check_losses(tag, job, ref_threads, losses, floor=FLOOR):
    mean, std, _ = band(job, ref_threads)
    tol = loss_tolerance(job, ref_threads, floor)
    dev = 100 * (losses - mean) / np.abs(mean)
    print("FIDELITY %s: losses %s | reference @%d threads %s (3 sigma %s %%) | deviation %s %% (allowed %s %%)" %
          (tag, np.round(losses).tolist(), ref_threads, np.round(mean).tolist(), np.round(300 * std / np.abs(mean), 2).tolist(),
           np.round(dev, 2).tolist(), np.round(100 * tol / np.abs(mean), 2).tolist()))
    assert len(losses) == len(mean)
    assert np.all(np.abs(losses - mean) <= tol), (tag, dev.tolist())


@pytest.fixture(scope="module")
def planted(tmp_path_factory):
    d = tmp_path_factory.mktemp("planted")
    corpus, questions = str(d / "planted.txt"), str(d / "questions.txt")
    ntok = make_planted(corpus, questions, repeats=120)
    assert ("%d tokens" % ntok) in BANDS["planted_b1_d200"]["corpus"]
    return corpus, questions, d


@pytest.mark.parametrize("job,threads", [("planted_b1_d200", 8), ("planted_b1_d200", 64),
                                         ("planted_cfg2_b2_d400", 8), ("planted_cfg2_b2_d400", 64)])
def test_planted_matches_reference_at_equal_thread_count(gpu, planted, job, threads):
    """epoch losses and total accuracy against the reference at the same thread count: the automatic kernel (plain) within
    FLOOR, the sentence-resident kernel (explicit choice) within FLOOR_RESIDENT; planted_cfg2_b2_d400 is BASELINE
    configs[2]'s shape (bitlevel 2, size 400, negative 24, iter 5) with its compute-accuracy parity"""
    corpus, questions, d = planted
    flags = BANDS[job]["flags"]
    _, _, acc_ref = band(job, threads)
    margin = max(ACC_EXCEPTION.get((job, threads), ACC_POINTS), 3 * float(acc_ref.std(ddof=1)))
    floor_auto = FLOOR_EXCEPTION.get((job, threads), FLOOR)
    floor_res = max(floor_auto, FLOOR_RESIDENT_EXCEPTION.get((job, threads), FLOOR_RESIDENT))
    for kernel, extra, floor in (("auto", [], floor_auto), ("resident", ["-window-cache", "1"], floor_res)):
        out = str(d / ("%s_%s_%d.bin" % (job, kernel, threads)))
        losses, _, _ = train(corpus, out, threads, flags, extra)
        acc = score(out, questions)
        assert acc["seen"] == acc["questions"] == 7728
        print("FIDELITY %s threads=%d %s: accuracy %.2f | reference %s +- %.1f" %
              (job, threads, kernel, acc["total"], acc_ref.tolist(), margin))
        check_losses("%s threads=%d %s" % (job, threads, kernel), job, threads, losses, floor)
        # (the sentence-resident kernel is an explicit choice with a looser loss bound; its accuracy is held to twice the margin)
        m = margin if kernel == "auto" else 2 * margin
        assert acc_ref.min() - m <= acc["total"] <= acc_ref.max() + m, (kernel, acc["total"], acc_ref.tolist())


def test_more_workers_than_the_corpus_supports_is_warned_about(gpu, planted):
    """512 workers on a 564 K-token corpus leaves 1 100 words per worker and epoch: alpha (re-computed per worker every
    10 000 words, ref :379-393) never moves.  The reference accepts that silently (its own 512-thread runs end 4 % off its
    8-thread ones, and diverge at bitlevel 2); `-threads 0` never picks such a count, and an explicit one is accepted with
    a warning.  The run itself is only bounded: bitlevel 1, plain kernel, epoch losses within 2 x FLOOR of the
    reference's 512-thread runs (measured in round 3: -1.3 % / +0.7 %)."""
    corpus, questions, d = planted
    flags = BANDS["planted_b1_d200"]["flags"]
    losses, _, err = train(corpus, str(d / "w512.bin"), 512, flags, ["-window-cache", "0"])
    assert "warning: -threads 512" in err and "picks at most" in err
    mean, _, _ = band("planted_b1_d200", 512)
    dev = (losses - mean) / np.abs(mean)
    print("FIDELITY planted_b1_d200 threads=512 plain: deviation %s %%" % np.round(100 * dev, 2).tolist())
    assert np.all(np.abs(dev) <= 2 * FLOOR)
    losses8, _, err8 = train(corpus, str(d / "w8.bin"), 8, dict(flags, iter=1), [])
    assert "warning" not in err8


@pytest.fixture(scope="module")
def text8size(tmp_path_factory):
    from w2b_testlib import write_zipf_text_corpus
    d = tmp_path_factory.mktemp("t8")
    return write_zipf_text_corpus(str(d / "c.txt")), d


@pytest.mark.parametrize("threads,ref_threads,kernels", [(64, 64, ("auto", "resident")), (256, 256, ("auto", "resident")),
                                                         (0, 256, ("auto",))])
def test_text8_size_matches_reference(gpu, text8size, threads, ref_threads, kernels):
    """17 M tokens, 70 K words, bitlevel 1, size 200, window 8, negative 24, 3 epochs: equal thread counts (64, 256), and
    `-threads 0` -- 256 workers here: 50 000 words per worker and epoch would be 340, and below a full device the library
    stays at the reference's own scale -- against the most threads the host can run at once"""
    corpus, d = text8size
    flags = BANDS["text8size"]["flags"]
    for kernel in kernels:
        extra = {"resident": ["-window-cache", "1"], "auto": []}[kernel]
        losses, workers, _ = train(corpus, str(d / "o.bin"), threads, flags, extra)
        check_losses("text8size threads=%d (%d workers) %s" % (threads, workers, kernel), "text8size", ref_threads, losses,
                     FLOOR if kernel == "auto" else FLOOR_RESIDENT)


def test_text8_size_window_residency_alone_is_loss_neutral(gpu, text8size):
    """same corpus, 128 workers, no hot-row copies, no lossless rows: what remains of the sentence-resident kernel (LDS window,
    scratch entries, exact-or-merge write-back) must give the plain kernel's epoch loss (measured: -58.58 M vs -58.62 M)"""
    corpus, d = text8size
    flags = dict(BANDS["text8size"]["flags"], iter=1)
    r, _, _ = train(corpus, str(d / "o.bin"), 128, flags, ["-window-cache", "1", "-hot-rows", "0", "-atomic-rank-u", "-1"])
    p, _, _ = train(corpus, str(d / "o.bin"), 128, flags, ["-window-cache", "0", "-hot-rows", "0", "-atomic-rank-u", "-1"])
    print("FIDELITY text8size threads=128 hot rows off: resident %s plain %s" % (r.tolist(), p.tolist()))
    assert np.all(np.abs(r - p) <= 0.005 * np.abs(p))


@pytest.fixture(scope="module")
def headline(tmp_path_factory):
    from w2b_testlib import write_headline_corpus
    d = tmp_path_factory.mktemp("hl")
    corpus = write_headline_corpus(str(d / "c.txt"))
    yield corpus, d
    os.remove(corpus)


@pytest.mark.parametrize("threads,ref_threads", [(0, 256), (1024, 256), (768, 256), (640, 256), (512, 256), (440, 256), (320, 256), (256, 256), (64, 64)])
def test_benchmarked_regime_matches_reference(gpu, headline, threads, ref_threads):
    """BASELINE configs[1] -- what bench.py times: V = 400 K, size 800, window 8, negative 24, bitlevel 1, -sample 0 -- on the
    22 M-token proxy file (the literal 100 M-token setting: test_benchmarked_setting_literally_100m_tokens).  `-threads 0` on
    this file is 440 workers (50 000 words per worker; not enough words for a full device: the mid range below); 1024 is what bench.py runs: a full device, per-XCD copies of
    the hottest rows (the CLI warns about the short shards; the reference's band is its 256-thread one, the most the host runs
    at once).  The row policy is a step function of the worker count and round 4 left the counts next to its steps untested;
    round 5 measured them (profiles/r05_sessions/r05m_gpu_runs.txt, r05q_mid_range.txt).  With every row shared the epoch
    loss drifts between the reference's scale and a full device: +1.0 / +1.3 / +1.6 / +1.5 / +0.7 % at 320 / 440 / 512 / 640 /
    767 workers.  From 257 to 640 workers four target rows therefore get per-XCD copies that are merged every word (-0.25 /
    -0.09 / -0.05 / -0.45 %; held-out 60 M-token regime: +0.18 -> -0.05 % at 440, +0.29 -> +0.03 % at 600); 641 ... 767 stay shared;
    from 768 the full-device copies (-0.6 ... -1.1 %).  All asserted here with the one 1.5 % floor."""
    corpus, d = headline
    flags = BANDS["headline"]["flags"]
    losses, workers, _ = train(corpus, "/dev/null", threads, flags)
    check_losses("headline threads=%d (%d workers)" % (threads, workers), "headline", ref_threads, losses)


@pytest.fixture(scope="module", params=sorted(HELDOUT))
def heldout(request, tmp_path_factory):
    from w2b_testlib import write_heldout_corpus
    d = tmp_path_factory.mktemp(request.param)
    corpus = write_heldout_corpus(str(d / "c.txt"), request.param)
    yield request.param, corpus
    os.remove(corpus)


@pytest.mark.parametrize("threads,ref_threads", [(0, 256), (256, 256), (64, 64)])
def test_held_out_regimes_match_reference(gpu, heldout, threads, ref_threads):
    """The regimes no knob was ever swept on (w2b_testlib.HELDOUT; bands recorded once, in round 4, before the rules that
    pass them were written): every rule that picks lossless rows / copies / the kernel from the word counts and the worker
    count has to carry over unchanged.  Same FLOOR as everywhere."""
    job, corpus = heldout
    flags = BANDS[job]["flags"]
    assert BANDS[job]["flags"] == HELDOUT[job]["flags"]
    losses, workers, _ = train(corpus, "/dev/null", threads, flags)
    check_losses("%s threads=%d (%d workers)" % (job, threads, workers), job, ref_threads, losses)


def test_full_device_on_a_held_out_regime(gpu, tmp_path):
    """The per-XCD copies of the hottest rows exist only when a launch fills the device, and their consensus rule is a balance
    measured on the benchmarked regime.  This is that setting on a regime it was never measured on: heldout_k5 on a 60 M-token
    stream, where `-threads 0` runs 1211 workers.  Band: the unmodified reference at 256 threads on the GPU box's host (5.4
    minutes per run; round 4 had ONE run, round 5 recorded another: the band is their mean, the tolerance max(3 sigma, FLOOR)).
    Measured: +0.6 % with the copies, +0.0 ... +0.8 % with `-hot-rows 0` (every row shared, lossless context rows) at the same
    1211 workers."""
    from w2b_testlib import write_heldout_corpus, HELDOUT_BIG
    job = "heldout_k5_big"
    corpus = write_heldout_corpus(str(tmp_path / "c.txt"), job)
    flags = BANDS[job]["flags"]
    assert flags == HELDOUT_BIG[job]["flags"]
    try:
        for extra in ([], ["-hot-rows", "0"]):
            losses, workers, _ = train(corpus, "/dev/null", 0, flags, extra)
            assert workers >= 768                                   # a full device: the copies are on in the default run
            check_losses("%s threads=0 (%d workers) %s" % (job, workers, " ".join(extra) or "default"), job, 256, losses)
    finally:
        os.remove(corpus)


def test_benchmarked_setting_literally_100m_tokens(gpu, tmp_path):
    """BASELINE configs[1] LITERALLY -- the 100 M-token stream bench.py times, at the bench's own 1024 workers (97 K words per
    worker) and as `-threads 0` picks them -- against the unmodified reference at 256 threads on the same file (job cfg1_100m:
    17 minutes of the GPU box's 256-thread host per run; round 5 recorded one run, round 6 a second: 0.08 % apart, so the
    tolerance is max(3 sigma, FLOOR) like everywhere else).
    An explicit `-threads 256` on this file is BETWEEN the reference's scale and a full device, where every row is shared and the
    epoch loss drifts on long streams (-3.6 % here, round 5: warned about, not gated).  Round 6: the command line then runs what
    `-threads 0` picks for the file, with a notice -- the reference's -threads is a speed knob, its results do not depend on it --
    and the run is GATED at the floor; `-threads-literal 1` keeps the count, warns, and is recorded."""
    from w2b_testlib import write_headline_corpus
    job = "cfg1_100m"
    corpus = write_headline_corpus(str(tmp_path / "c.txt"), n_zipf=98_000_000)
    flags = BANDS[job]["flags"]
    try:
        for threads in (1024, 0, 256):
            losses, workers, err = train(corpus, "/dev/null", threads, flags)
            assert workers >= 768 and "warning" not in err
            assert ("notice: -threads 256" in err) == (threads == 256)
            check_losses("%s threads=%d (%d workers)" % (job, threads, workers), job, 256, losses)
        losses, workers, err = train(corpus, "/dev/null", 256, flags, ["-threads-literal", "1"])
        mean, _, _ = band(job, 256)
        dev = (losses - mean) / np.abs(mean)
        print("FIDELITY %s threads=256 -threads-literal 1 (%d workers, shared rows; NOT a gate): deviation %s %%" % (job, workers, np.round(100 * dev, 2).tolist()))
        assert workers == 256 and "drifts on long streams" in err and np.all(np.abs(dev) <= 0.08)
    finally:
        os.remove(corpus)


def test_full_device_on_a_second_held_out_long_stream(gpu, tmp_path):
    """heldout_v1m (round 6; asked for by the round-5 review): V = 1 M, Zipf exponent 1.1, size 512, window 5, negative 10, 85 M
    words -- a second full-device long-stream regime, its reference band recorded BEFORE any constant of the full-device mode
    (merge period 16 -- tuned on the very stream it was asserted on --, weight 1/8, ~113 + 113 copies) was touched again.
    `-threads 0` runs 1701 workers with per-XCD copies.  Band: two 256-thread runs of the unmodified reference (0.37 % apart,
    sigma 0.26 % -- this regime is the reference's least repeatable one).  Measured with the shipped defaults (profiles/r06_sessions/
    r06b_fidelity_runs.txt, r06e, r06j, r06m): seven runs between -336.4 and -337.3 M = +1.15 ... +1.42 % of the band's mean (mean
    +1.29 %, sigma 0.10 %); `-hot-rows 0`: -1.0 %.  This is the regime closest to the floor; the gate is on the MEAN of four runs
    (a band is a mean too), so that the product's own scatter does not decide it."""
    from w2b_testlib import write_heldout_corpus, HELDOUT_BIG
    job = "heldout_v1m"
    corpus = write_heldout_corpus(str(tmp_path / "c.txt"), job)
    flags = BANDS[job]["flags"]
    assert flags == HELDOUT_BIG[job]["flags"]
    try:
        runs = []
        for _ in range(4):
            losses, workers, _ = train(corpus, "/dev/null", 0, flags)
            assert workers >= 768
            runs.append(losses)
        check_losses("%s threads=0 (%d workers), mean of four runs %s" % (job, workers, [r.tolist() for r in runs]), job, 256, np.mean(runs, 0))
    finally:
        os.remove(corpus)


def test_long_streams_at_short_rows(gpu, tmp_path):
    """(round-5 review, weak 1d) the row lengths at which the row-group kernel is automatic -- BASELINE configs[0] (size 200) and
    configs[2] (size 400, 2 bits) -- on a 100 M-token stream over the text8-sized vocabulary (default sub-sampling): explicit
    `-threads 256` (row groups) and `-threads 0` (a full device here: plain kernel with copies) against two 256-thread runs of the
    unmodified reference each (sigma 0.01 %).  Measured (r06b): size 200 +0.11 / +0.65 %, size 400 / 2 bits -0.62 / +0.92 %."""
    from w2b_testlib import write_heldout_corpus, HELDOUT_BIG
    corpus = write_heldout_corpus(str(tmp_path / "c.txt"), "long_d200")
    try:
        for job in ("long_d200", "long_d400b2"):
            flags = BANDS[job]["flags"]
            assert flags == HELDOUT_BIG[job]["flags"] and HELDOUT_BIG[job]["corpus"] == HELDOUT_BIG["long_d200"]["corpus"]
            for threads in (256, 0):
                losses, workers, err = train(corpus, "/dev/null", threads, flags)
                assert "notice" not in err                      # (an explicit count that the row-group kernel runs is kept)
                check_losses("%s threads=%d (%d workers)" % (job, threads, workers), job, 256, losses)
    finally:
        os.remove(corpus)
