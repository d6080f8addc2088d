#!/usr/bin/env python3
"""bench.py -- training words/sec of the Word2Bits hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[1]): synthetic 100M-token Zipf(1) stream, vocab=400K, bitlevel=1,
size=800, window=8, negative=24, sample=0, on one MI355X.  A "step" is ONE launch of the fused
CBOW/negative-sampling update kernel over one batch of centre words:

  --form worker  (default) : every Hogwild worker (workgroup) advances --positions sentence positions
                             of the resident token stream, drawing windows/negatives on device exactly
                             like TrainModelThread (ref src/word2bits.cpp:363-516); 2^20 centre words per step
  --form tuples            : 2^20 explicit (centre, 9 context rows, 24 negatives) tuples per step,
                             SURVEY.md 8d / north_star "synthetic (center, context, K-negatives) tuples"
  --form eval              : NOT the headline metric -- the analogy evaluator's scan (include/word2bits_eval.h,
                             SURVEY.md 8 f4) with its own metric, roofline (fp32 MFMA) and CPU reference

All inputs are resident in HBM before the timed region.  N>1: one process per GPU
(torch.distributed.run), each rank trains its own shard (weak scaling) on its own replica and
the replicas are combined every --sync-every steps by an RCCL all-reduce of [u||v] (--sync-mode,
default 2: saturation + quantization cells), inside the timed region.  `--gpus N` launched without a rendezvous
starts the N ranks itself (torch.distributed.run); with a rendezvous of another size it exits 2.

The headline runs WITH the loss bookkeeping (--loss 1): the instantiation ./word2bits runs (it prints "Epoch Loss").

Prints ONE JSON line on rank 0 (contract in the task description) with extra objects:
  roofline     -- algorithmic HBM bytes per launch / hipEvent-measured launch duration vs 8 TB/s, min / median / max per
                  launch, and the HBM bytes of the calibrated rocprofv3 counters (quoted from profiles/r06_pmc_worker.json:
                  traffic_measured_in_this_run = false)
  cpu_baseline -- the reference CPU program (oracle/_ref/word2bits_stock, built from the unmodified reference sources)
                  on all host threads, training phase of a whole epoch over a bounded corpus of the same shape;
                  cpu_baseline_1thread (the same with -threads 1, a 20 s sample), cpu_baseline_configs0 (BASELINE
                  configs[0]: size 200, -threads 1)
  legs         -- without_loss_bookkeeping, bitlevel2, relaxed_coherence, other_shapes (tuples; configs[4] shape at bitlevel 1 and 0;
                  size 200 and size 400 / 2 bits with the automatic kernel -- the row-group kernel since round 5 -- and with the
                  plain kernel beside it; partial_device = the headline shape on a 22 M-token stream; the explicit
                  sentence-resident kernel), e2e (./word2bits end to end, measured in this run; the reference's side quoted
                  from profiles/), us_per_word_per_worker
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12   # B/s, /opt/skills/guides/MI355X_MICROARCH.md (spec peak)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--form", choices=["tuples", "worker", "eval"], default="worker",
                    help="worker / tuples: the training hot path (the headline metric); eval: the analogy evaluator's "
                         "scan (include/word2bits_eval.h), reported with its own metric")
    ap.add_argument("--vocab", type=int, default=400_000)
    ap.add_argument("--dim", type=int, default=800)
    ap.add_argument("--window", type=int, default=8)
    ap.add_argument("--negative", type=int, default=24)
    ap.add_argument("--bitlevel", type=int, default=1)
    ap.add_argument("--tokens", type=int, default=100_000_000)
    ap.add_argument("--batch", type=int, default=1 << 20, help="centre words per step (tuples form)")
    ap.add_argument("--workers", type=int, default=0,
                    help="Hogwild workers (worker form); 0 = exactly the workgroups resident on the GPU")
    ap.add_argument("--positions", type=int, default=0,
                    help="sentence positions per worker per step (0: --batch / workers)")
    ap.add_argument("--ids", choices=["zipf", "uniform"], default="zipf")
    ap.add_argument("--zipf-shift", type=int, default=0,
                    help="experiment: Zipf weights 1/(rank + N) instead of 1/rank (a Zipf stream without its N hottest words)")
    ap.add_argument("--sync-every", type=int, default=1,
                    help="N>1: steps between two full replica exchanges (1 = after every step, what ./word2bits -gpus N does: the "
                         "interval is what costs epoch loss, tests/test_gpu_exchange.py; 16 until round 4)")
    ap.add_argument("--sync-mode", type=int, default=2,
                    help="0 delta-sum, 1 average, 2 saturation per row + quantization cells per element (what ./word2bits -gpus N uses)")
    ap.add_argument("--sync-impl", choices=["lib", "torch"], default="lib",
                    help="replica exchange: lib = the library's own RCCL communicator (w2b_comm_init / "
                         "w2b_sync_replicas: what the CLI uses; falls back to torch if its initialisation fails on any "
                         "rank); torch = torch.distributed all_reduce (RCCL) on a zero-copy view of the library's "
                         "[u||v] buffer")
    ap.add_argument("--cpu-baseline", choices=["reference", "port", "none"], default="reference")
    ap.add_argument("--cpu-tokens", type=int, default=6_000_000)
    ap.add_argument("--cpu-1thread-seconds", type=float, default=20.0, help="sample length of the -threads 1 reference leg")
    ap.add_argument("--cpu-cfg0", type=int, default=1, help="1: also time BASELINE configs[0] (the reference's own CPU case, -threads 1)")
    ap.add_argument("--also-relaxed", type=int, default=1,
                    help="N=1 only: after the headline (coherent) run, time the same steps with relaxed row "
                         "coherence and report it as an extra object")
    ap.add_argument("--also-legs", type=int, default=1,
                    help="N=1, worker form only: after the headline, time the same steps (a) with the loss bookkeeping "
                         "on (the instantiation ./word2bits runs: it prints 'Epoch Loss') and (b) at bitlevel 2, and "
                         "report both as extra objects")
    ap.add_argument("--also-shapes", type=int, default=1,
                    help="N=1, default workload only: after the headline, run the other measured shapes (form (ii) tuples, "
                         "BASELINE configs[4] shape at bitlevel 1 and 0, configs[0] / configs[2] row lengths) as short "
                         "sub-runs of this script and report each with its own value, roofline fraction and kernel")
    ap.add_argument("--loss", type=int, default=1,
                    help="1 (default): the headline runs with the loss bookkeeping on -- the instantiation ./word2bits runs (it "
                         "always prints 'Epoch Loss', ref :539); 0: without it")
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--window-cache", type=int, default=-1,
                    help="worker form: -1 = automatic, 1 = sentence-resident kernel (context window rows stay in "
                         "LDS), 0 = plain kernel")
    ap.add_argument("--row-groups", type=int, default=-1,
                    help="worker form: -1 = automatic, 1 = the row-group worker kernel wherever it fits, 0 = never")
    ap.add_argument("--refresh-rows", type=int, default=0, help="w2b_tuning.refresh_rows_u (0 = library default, -1 = none)")
    ap.add_argument("--relaxed", type=int, default=0,
                    help="1: plain cached row accesses (not coherent between XCDs); default 0 = agent-scope (sc1)")
    ap.add_argument("--hot-rows", type=int, default=-1,
                    help="w2b_tuning.hot_rows_v / hot_rows_u: leading rows with per-XCD copies (-1 = from the word counts)")
    ap.add_argument("--hot-period", type=int, default=0, help="w2b_tuning.hot_period (0 = library default)")
    ap.add_argument("--hot-cap", type=int, default=-1, help="w2b_tuning.hot_cap (-1 = library default)")
    ap.add_argument("--atomic-rank", type=int, default=-2, help="w2b_tuning.atomic_rank (-2 = library default, -1 = automatic)")
    ap.add_argument("--atomic-cap", type=int, default=-1, help="w2b_tuning.atomic_cap (-1 = library default)")
    ap.add_argument("--hot-weight", type=int, default=0, help="w2b_tuning.hot_weight_permille (0 = library default)")
    ap.add_argument("--atomic-rank-u", type=int, default=0, help="w2b_tuning.atomic_rank_u (0 = as atomic_rank, -1 = none)")
    ap.add_argument("--fresh-rank-u", type=int, default=0, help="w2b_tuning.fresh_rank_u (0 = library default, -1 = none)")
    ap.add_argument("--window-refresh", type=int, default=-1, help="w2b_tuning.window_refresh (-1 = library default)")
    ap.add_argument("--eval-questions", type=int, default=19544, help="--form eval: questions (questions-words.txt)")
    ap.add_argument("--eval-kind", choices=["1bit", "fp"], default="1bit")
    ap.add_argument("--eval-cpu-questions", type=int, default=24)
    return ap.parse_args()


def latest_profile(name):
    """profiles/rNN_<name> of the newest round that has it (committed measurements that this script quotes, never re-labels)"""
    for rnd in ("r06", "r05", "r04", "r03"):
        f = os.path.join(ROOT, "profiles", "%s_%s" % (rnd, name))
        if os.path.exists(f):
            return f
    return os.path.join(ROOT, "profiles", "r05_%s" % name)


def algorithmic_bytes_per_word(D, cw, K):
    # every touched row read once + written once, ids read once
    return 8 * D * (cw + K + 1) + 4 * (1 + cw + K)


def workload_name(args):
    """which BASELINE.json configuration (if any) the arguments describe"""
    shape = (args.vocab, args.dim, args.window, args.negative)
    if shape == (400_000, 800, 8, 24) and args.bitlevel == 1 and args.ids == "zipf":
        return "BASELINE configs[1]" if args.tokens == 100_000_000 else "BASELINE configs[1] shape (%d tokens)" % args.tokens
    if shape == (3_700_000, 1000, 8, 12) and args.bitlevel in (0, 1):
        return "BASELINE configs[4] shape (one GPU of the eight)"
    if args.dim == 200 and args.bitlevel == 1 and args.negative == 24 and args.window == 8:
        return "BASELINE configs[0] shape (size 200, bitlevel 1) on a synthetic vocabulary"
    if args.dim == 400 and args.bitlevel == 2 and args.negative == 24 and args.window == 8:
        return "BASELINE configs[2] shape (size 400, bitlevel 2) on a synthetic vocabulary"
    return "custom shape"


def zipf_cdf(torch, V, device, uniform, shift=0):
    if uniform:
        w = torch.ones(V - 1, dtype=torch.float64, device=device)
    else:
        w = 1.0 / (torch.arange(1, V, dtype=torch.float64, device=device) + shift)
    cdf = torch.cumsum(w, 0)
    return cdf / cdf[-1]


def draw_ids(torch, cdf, n, gen):
    out = torch.empty(n, dtype=torch.int32, device=cdf.device)
    chunk = 1 << 24
    for o in range(0, n, chunk):
        m = min(chunk, n - o)
        r = torch.rand(m, dtype=torch.float64, device=cdf.device, generator=gen)
        out[o:o + m] = (torch.searchsorted(cdf, r) + 1).clamp_(max=len(cdf)).to(torch.int32)
    return out


# ------------------------------------------------------------------------------------------ CPU baseline
class RefProbe:
    """The UNMODIFIED reference program (oracle/_ref/word2bits_stock) on this host, watched through its stdout (stdbuf -o0:
    unbuffered, so the marker lines arrive when they are printed; no pty needed): 'Starting epoch: 0' -> 'Epoch Loss:' is
    the training phase (never "total minus -iter 0"), and the progress line's percentage gives the words done
    at any moment for a bounded sample of a run that would take too long."""

    def __init__(self, path, flags, threads, sample_seconds=None, timeout=600):
        import threading
        self.exe = os.path.join(ROOT, "oracle", "_ref", "word2bits_stock")
        self.threads, self.sample_seconds, self.timeout = threads, sample_seconds, timeout
        self.t_launch = time.time()
        self.t_start = self.t_end = self.words = None
        self.progress = []                        # (seconds since 'Starting epoch', percent)
        cmd = ["stdbuf", "-o0", self.exe, "-train", path, "-output", "/dev/null", "-threads", str(threads)] + flags
        self.p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, bufsize=0)
        self.th = threading.Thread(target=self._watch, daemon=True)
        self.th.start()

    def _watch(self):
        import re
        import select
        fd = self.p.stdout.fileno()
        buf = b""
        deadline = time.time() + self.timeout
        while time.time() < deadline:
            r, _, _ = select.select([fd], [], [], 0.5)
            now = time.time()
            if r:
                chunk = os.read(fd, 65536)
                if not chunk:
                    break
                buf += chunk
                if self.words is None:
                    m = re.search(rb"Words in train file: (\d+)", buf)
                    if m:
                        self.words = int(m.group(1))
                if self.t_start is None and b"Starting epoch: 0" in buf:
                    self.t_start = now
                if self.t_start is not None:
                    m = re.findall(rb"Progress: ([\d.]+)%", buf[-400:])
                    if m:
                        self.progress.append((now - self.t_start, float(m[-1])))
                if b"Epoch Loss:" in buf:
                    self.t_end = now
                    break
                if len(buf) > (1 << 20):
                    buf = buf[-4096:]
            elif self.p.poll() is not None:
                break
            if self.sample_seconds and self.t_start is not None and now - self.t_start >= self.sample_seconds:
                break
        self.p.kill()                            # the save loop that follows is not part of the metric
        self.p.wait()

    def result(self):
        self.th.join(self.timeout + 5)
        if self.t_start is None or not self.words:
            return None
        startup = self.t_start - self.t_launch
        if self.t_end is not None:               # a whole epoch
            return {"words_per_s": self.words / (self.t_end - self.t_start), "train_s": self.t_end - self.t_start,
                    "startup_s": startup, "words": self.words, "whole_epoch": True}
        pr = [(t, p) for t, p in self.progress if t >= 1.0]
        if len(pr) < 2 or pr[-1][0] - pr[0][0] < 2.0:
            return None
        dw = (pr[-1][1] - pr[0][1]) / 100.0 * (self.words + 1)     # Progress = wca / (iter * train_words + 1), ref :385
        return {"words_per_s": dw / (pr[-1][0] - pr[0][0]), "train_s": pr[-1][0] - pr[0][0], "startup_s": startup,
                "words": int(dw), "whole_epoch": False}


def write_cpu_corpus(args):
    """Corpus per BASELINE.md section 4 / SURVEY Appendix C.8: every vocabulary word 5x (so that -min-count 5 keeps V
    rows), then a Zipf(1) stream; newline every 1000 tokens."""
    import tempfile
    V, nz = args.vocab, args.cpu_tokens
    rng = np.random.default_rng(1234)
    base = np.repeat(np.arange(1, V, dtype=np.int64), 5)
    rng.shuffle(base)
    w = 1.0 / np.arange(1, V, dtype=np.float64)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    z = np.searchsorted(cdf, rng.random(nz)) + 1
    ids = np.concatenate([base, z])
    tmpdir = tempfile.mkdtemp(prefix="w2b_cpu_")
    path = os.path.join(tmpdir, "corpus.txt")
    words = np.array([b"w%d" % i for i in range(V)], dtype=object)
    with open(path, "wb") as f:
        for o in range(0, len(ids), 1000):
            f.write(b" ".join(words[ids[o:o + 1000]]))
            f.write(b"\n")
    return path


def ref_flags(args):
    return ["-bitlevel", str(args.bitlevel), "-size", str(args.dim), "-window", str(args.window), "-negative",
            str(args.negative), "-iter", "1", "-sample", "0", "-binary", "1", "-min-count", "5"]


def cpu_baseline_reference(args, path):
    """the headline shape on ALL host threads, training phase of a whole epoch over the bounded corpus"""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "word2bits_stock")):
        return None
    cores = os.cpu_count() or 1
    r = RefProbe(path, ref_flags(args), cores).result()
    if not r:
        return None
    return {"value": r["words_per_s"], "unit": "words/s", "cores": cores, "kind": "reference",
            "sample": "unmodified reference CPU program (oracle/_ref/word2bits_stock: -O3 -march=x86-64-v3, "
                      "FMA contraction on), %d tokens = every one of %d words 5x + %d Zipf(1) tokens, V=%d "
                      "D=%d w=%d K=%d bitlevel=%d -sample 0 -threads %d; training phase only "
                      "('Starting epoch' -> 'Epoch Loss'), %.1f s (start-up before it: %.1f s)"
                      % (r["words"], args.vocab - 1, args.cpu_tokens, args.vocab, args.dim, args.window, args.negative,
                         args.bitlevel, cores, r["train_s"], r["startup_s"])}


def cpu_baseline_cfg0():
    """BASELINE configs[0] literally (the reference's own CPU-runnable case): bitlevel 1, size 200, window 8, negative 24,
    iter 1, -threads 1 -- on the planted-analogy corpus (text8 is not available offline; 645 K words)."""
    import tempfile
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "word2bits_stock")):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from planted import make_planted
    d = tempfile.mkdtemp(prefix="w2b_cfg0_")
    corpus, questions = os.path.join(d, "planted.txt"), os.path.join(d, "q.txt")
    make_planted(corpus, questions, repeats=120)
    r = RefProbe(corpus, ["-bitlevel", "1", "-size", "200", "-window", "8", "-negative", "24", "-iter", "1", "-binary", "1",
                          "-min-count", "5"], 1, timeout=300).result()
    for f in (corpus, questions):
        try:
            os.remove(f)
        except OSError:
            pass
    if not r:
        return None
    return {"value": r["words_per_s"], "unit": "words/s", "cores": 1, "kind": "reference",
            "sample": "BASELINE configs[0]: ./word2bits -bitlevel 1 -size 200 -window 8 -negative 24 -iter 1 -threads 1 (default "
                      "-sample) on the planted-analogy corpus standing in for text8: %d words, whole epoch %.1f s" % (r["words"], r["train_s"])}


def cpu_baseline_port(args):
    """Time the oracle restatement (oracle/libw2b_oracle.so), Hogwild over all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from w2b_testlib import OracleState
    V, D = args.vocab, args.dim
    rng = np.random.default_rng(7)
    n = args.cpu_tokens
    w = 1.0 / np.arange(1, V, dtype=np.float64)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    ids = (np.searchsorted(cdf, rng.random(n)) + 1).astype(np.int32)
    ids[999::1000] = 0
    cn = np.bincount(ids, minlength=V).astype(np.int64)
    cn[cn == 0] = 1
    cores = os.cpu_count() or 1
    o = OracleState(cn, D, window=args.window, negative=args.negative, bitlevel=args.bitlevel,
                    num_threads=cores, iters=1, sample=0.0, table_size=100_000_000, compute_loss=0)
    starts = (np.arange(cores, dtype=np.int64) * (n // cores))
    t0 = time.time()
    o.train_epoch_tokens(ids, starts)
    dt = time.time() - t0
    return {"value": n / dt, "unit": "words/s", "cores": cores, "kind": "port",
            "sample": "oracle/w2b_oracle.c (bit-exact restatement, -O2 no FMA), %d Zipf(1) tokens, V=%d D=%d "
                      "w=%d K=%d bitlevel=%d, %d pthreads Hogwild, %.1f s" %
                      (n, V, D, args.window, args.negative, args.bitlevel, cores, dt)}


# ------------------------------------------------------------------------------------------ --form eval
def run_eval_form(args):
    """The evaluator's scan: `steps` batched top-1 scans of --eval-questions questions over a
    --vocab x --dim matrix resident in HBM (defaults of this form: text8-sized 60238 x 200, 1-bit).  One JSON line
    in the shape of the contract; the unmodified reference evaluator (oracle/_ref/compute_accuracy) is timed
    beside it on a bounded sample as `cpu_baseline`."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import word2bits_amd as w2b
    from w2b_testlib import write_vectors_file, ref_binary
    V = args.vocab if args.vocab != 400_000 else 60238      # text8' vocabulary (reference README.md:122-131)
    D = args.dim if args.dim != 800 else 200
    Q, steps, warmup = args.eval_questions, args.steps, max(1, args.warmup)
    rng = np.random.default_rng(3)
    M = ((rng.integers(0, 2, (V, D)) * 2 - 1).astype(np.float32) / np.float32(3)) if args.eval_kind == "1bit" \
        else rng.standard_normal((V, D)).astype(np.float32)
    path = write_vectors_file(os.path.join(tempfile.mkdtemp(), "v.bin"), [("w%d" % i).encode() for i in range(V)], M)
    b = rng.integers(0, V, (3, Q)).astype(np.int32)
    peak = {True: 78.6e12, False: 39.3e12}    # multiply-adds/s: 256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz / 2 (fused), /4
    modes, answers = {}, {}
    for fused in (True, False):
        ev = w2b.Evaluator(path, 0, 0, fused=fused)
        for _ in range(warmup):
            ev.top1(*b)
        ev.timing()
        t0 = time.perf_counter()
        for _ in range(steps):
            answers[fused], _ = ev.top1(*b)
        wall = (time.perf_counter() - t0) / steps
        ms, launches, macs = ev.timing()
        rate = macs / (ms * 1e-3)
        modes[fused] = {"questions_per_s": Q / wall, "ms_per_step": wall * 1e3, "kernel_ms": ms / launches,
                        "achieved": 2 * rate / 1e12, "peak": 2 * peak[fused] / 1e12, "frac": rate / peak[fused]}
        ev.close()
    f = modes[True]
    result = {
        "metric": "analogy questions/sec (exhaustive top-1 scan, vocab=%d dim=%d)" % (V, D),
        "value": f["questions_per_s"], "unit": "questions/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": f["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "analogy scan: %d questions x %d rows x %d dims (%s vectors), arithmetic of the "
                               "reference's stock (FMA) build" % (Q, V, D, args.eval_kind)},
        "roofline": {"bound": "mfma", "achieved": f["achieved"], "peak": f["peak"], "unit": "TFLOP/s",
                     "frac": f["frac"], "traffic": None, "kernel": "k_eval_scores_mfma", "avg_launch_ms": f["kernel_ms"]},
        "two_rounding_mode": {"kernel": "k_eval_scores<false> (packed fp32 VALU)", **modes[False]},
        "answers_differ_between_modes": int((answers[True] != answers[False]).sum()),
    }
    exe = ref_binary("compute_accuracy")
    n = args.eval_cpu_questions
    if exe and n > 0 and args.cpu_baseline != "none":
        qs = ": s\n" + "".join("w%d w%d w%d w%d\n" % (b[0, i], b[1, i], b[2, i], b[2, i]) for i in range(n))
        t0 = time.perf_counter()
        subprocess.run([exe, path, "0", "0"], input=b"", capture_output=True)
        t_load = time.perf_counter() - t0
        t0 = time.perf_counter()
        subprocess.run([exe, path, "0", "0"], input=qs.encode(), capture_output=True)
        t_all = time.perf_counter() - t0
        result["cpu_baseline"] = {"value": n / max(t_all - t_load, 1e-9), "unit": "questions/s", "cores": 1,
                                  "kind": "reference",
                                  "sample": "%d questions through the unmodified evaluator (single-threaded program); "
                                            "its load time (%.1f s) subtracted" % (n, t_load)}
    else:
        result["cpu_baseline"] = None
    print(json.dumps(result), flush=True)


def other_shapes():
    """the other shapes DESIGN.md section 6 quotes, each as a short sub-run of this script (fresh process, same code path),
    so that the driver's line carries them: value, roofline fraction (algorithmic bytes of THAT shape), kernel, workload"""
    legs = {
        "tuples": ["--form", "tuples"],
        # (configs[4] is a Wikipedia-scale stream sharded over 8 GPUs: a full device per GPU, which -threads 0 gives from
        # 52 M tokens on -- 60 M here; the text8-shaped legs below keep a stream that leaves the device partly filled)
        "cfg5_b1": ["--vocab", "3700000", "--dim", "1000", "--negative", "12", "--bitlevel", "1", "--tokens", "60000000"],
        "cfg5_b0": ["--vocab", "3700000", "--dim", "1000", "--negative", "12", "--bitlevel", "0", "--tokens", "60000000"],
        # text8-shaped legs: a 30 M-token stream leaves the device partly filled (-threads 0 = 256 workers); automatic = the
        # row-group kernel since round 5 (every row shared, lossless context rows, the very hottest context rows read at
        # refreshed copies); the plain kernel (round 4's automatic choice) beside it
        "d200": ["--vocab", "60238", "--dim", "200"],
        "d400_b2": ["--vocab", "60238", "--dim", "400", "--bitlevel", "2"],
        "d200_plain": ["--vocab", "60238", "--dim", "200", "--row-groups", "0"],
        "d400_b2_plain": ["--vocab", "60238", "--dim", "400", "--bitlevel", "2", "--row-groups", "0"],
        # what a user of ./word2bits -threads 0 gets on a file too short for a full device: the headline shape on a 22 M-token stream
        "partial_device": ["--tokens", "22000000"],
        # the sentence-resident kernel (context window rows stay in LDS): an explicit choice since round 4 (./word2bits
        # -window-cache 1) because it keeps context rows private for up to 2 x window + 1 positions and is up to 13 % off the
        # reference's epoch loss on a held-out regime
        "cfg5_b1_resident": ["--vocab", "3700000", "--dim", "1000", "--negative", "12", "--bitlevel", "1", "--window-cache", "1", "--tokens", "60000000"],
        "d200_resident": ["--vocab", "60238", "--dim", "200", "--window-cache", "1"],
    }
    # tables of 48 / 96 MB live in the 256 MB Infinity Cache: HBM's 8 TB/s is not what bounds those legs.  The bound quoted
    # beside it is what the memory system sustains for the same access shape (random rows, 16 bytes per lane, sc1 read +
    # write) on a cache-sized table: tools/row_probe small, measured in the round's profile session
    cache_bound = {}
    try:
        cache_bound = json.load(open(latest_profile("cache_bound.json")))
    except Exception:
        pass
    out = {}
    for name, extra in legs.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--tokens", "30000000", "--steps", "10", "--warmup", "3",
               "--cpu-baseline", "none", "--also-relaxed", "0", "--also-legs", "0", "--also-shapes", "0"] + extra   # (a later --tokens wins)
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            d = json.loads(line[-1])
            out[name] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                         "roofline_frac": d["roofline"]["frac"], "achieved_GBps": d["roofline"]["achieved"],
                         "kernel": d["roofline"]["kernel"], "avg_launch_ms": d["roofline"]["avg_launch_ms"],
                         "workload": d["config"]["workload"], "worker_kernel": d["config"].get("worker_kernel"),
                         "workers": d["config"].get("workers"), "us_per_word_per_worker": d.get("us_per_word_per_worker")}
            cb = cache_bound.get(name.replace("_resident", ""))
            if cb:
                out[name]["cache_resident_bound_GBps"] = cb["GBps"]
                out[name]["frac_of_cache_resident_bound"] = d["roofline"]["achieved"] / cb["GBps"]
        except Exception as e:                  # a leg must never take the headline down
            out[name] = {"value": None, "error": repr(e)[:200]}
    return out


# ------------------------------------------------------------------------------------------ ranks
def ensure_ranks(args):
    """`--gpus N` means N ranks, one per GPU -- or no line at all.

    * launched bare (no WORLD_SIZE in the environment) with N > 1: this process re-executes itself under
      `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (what the driver does
      itself for N > 1) and exits with that job's status;
    * launched by a rendezvous whose WORLD_SIZE differs from --gpus: exit status 2, nothing printed on stdout -- a
      line that says `n_gpus: 1` for a `--gpus 8` command (round 3: --gpus was parsed and never read) cannot happen.
    Returns (world, rank, local_rank)."""
    ws = os.environ.get("WORLD_SIZE")
    if ws is None:
        if args.gpus <= 1:
            return 1, 0, 0
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print("bench.py: --gpus %d without a rendezvous: starting %d ranks (%s)" % (args.gpus, args.gpus, " ".join(cmd[1:9])),
              file=sys.stderr, flush=True)
        raise SystemExit(subprocess.call(cmd))
    world = int(ws)
    if world != args.gpus:
        print("bench.py: --gpus %d but the rendezvous has WORLD_SIZE=%d ranks; refusing to report a line for a job "
              "that is not the one asked for" % (args.gpus, world), file=sys.stderr, flush=True)
        raise SystemExit(2)
    return world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def dry_run(args, world, rank):
    """W2B_BENCH_DRY=1 (test hook, CPU): everything of the N-rank protocol that needs no GPU -- rendezvous (gloo),
    barrier + max-over-ranks timing of K empty steps, ONE line from rank 0 -- so that `bench.py --gpus 2`, launched
    bare, can be checked to produce two ranks on a machine without a GPU.  The line says so and carries no value."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        assert dist.get_world_size() == world
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ranks_seen = 1
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        one = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(one)
        dt, ranks_seen = float(tt.item()), int(one.item())
    if rank == 0:
        print(json.dumps({"metric": "training words/sec at dim=%d bitlevel=%d neg=%d" % (args.dim, args.bitlevel, args.negative),
                          "value": None, "unit": "words/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / max(1, args.steps) * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "none",
                          "dry_run": "W2B_BENCH_DRY=1: rendezvous and timing protocol only, no GPU work -- not a measurement",
                          "ranks_seen": ranks_seen, "rccl_ranks": 0}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------ main
def main():
    args = parse()
    if args.form == "eval":
        return run_eval_form(args)
    world, rank, local_rank = ensure_ranks(args)
    if os.environ.get("W2B_BENCH_DRY") == "1":
        return dry_run(args, world, rank)
    import torch
    import torch.distributed as dist
    import word2bits_amd as w2b

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if world > torch.cuda.device_count() and os.environ.get("W2B_BENCH_SHARE_GPU") != "1":
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible (one rank per GPU; W2B_BENCH_SHARE_GPU=1 is the "
                         "smoke-test hook that lets ranks share devices)" % (world, torch.cuda.device_count()))
    # (test hook for 1-GPU boxes: W2B_BENCH_SHARE_GPU=1 puts every rank on the devices that exist and
    # W2B_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one device -- the N > 1 control flow of this
    # file can then be smoke-tested; such a line says so in `config.replica_sync` and is not a measurement)
    shared = os.environ.get("W2B_BENCH_SHARE_GPU") == "1"
    if shared:
        local_rank %= torch.cuda.device_count()
    backend = os.environ.get("W2B_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    V, D, W, K = args.vocab, args.dim, args.window, args.negative
    cw = W + 1                                   # mean context words at window W
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)                 # stream seed = rank (mirrors ref :368)

    # ---- synthetic corpus resident in HBM (untimed)
    per_rank_tokens = args.tokens                 # weak scaling: the same stream length per GPU
    cdf = zipf_cdf(torch, V, dev, args.ids == "uniform", args.zipf_shift)
    stream = draw_ids(torch, cdf, per_rank_tokens, gen)
    stream[999::1000] = 0                        # "</s>" every 1000 tokens
    counts_t = torch.bincount(stream.long(), minlength=V)
    if world > 1:                                 # one vocabulary for all replicas: global word counts
        dist.all_reduce(counts_t)
    counts = counts_t.clamp_(min=1).cpu().numpy().astype(np.int64)
    train_words = int(counts.sum()) // world      # per-rank share; the trainer gets the global number below

    props = torch.cuda.get_device_properties(dev)
    ncu = props.multi_processor_count
    wcache = None if args.window_cache < 0 else bool(args.window_cache)
    rgroups = None if args.row_groups < 0 else bool(args.row_groups)
    workers = args.workers
    if workers <= 0:          # ask the library how many workgroups of the worker kernel are resident at once
        probe = w2b.Trainer(V, D, W, K, args.bitlevel, num_threads=1, device=local_rank, sample=0.0,
                            train_words=train_words * world,
                            relaxed_coherence=bool(args.relaxed), window_cache=wcache, row_groups=rgroups, compute_loss=bool(args.loss))
        probe.set_vocab_counts(counts, 0)         # which kernel runs (and how many workers fill the device) depends on the counts
        workers = probe.suggested_threads()
        probe.close()
    from word2bits_amd import replicas
    nw_local = workers if args.form == "worker" else 1
    worker_offset, _ = replicas.worker_plan(nw_local * world, world, rank)   # global Hogwild worker ids

    tune = {}
    if args.hot_rows >= 0:
        tune.update(hot_rows_v=args.hot_rows, hot_rows_u=args.hot_rows)
    if args.hot_period > 0:
        tune["hot_period"] = args.hot_period
    if args.hot_cap >= 0:
        tune["hot_cap"] = args.hot_cap
    if args.atomic_rank >= -1:
        tune["atomic_rank"] = args.atomic_rank
    if args.atomic_cap >= 0:
        tune["atomic_cap"] = args.atomic_cap
    if args.hot_weight > 0:
        tune["hot_weight_permille"] = args.hot_weight
    if args.window_refresh >= 0:
        tune["window_refresh"] = args.window_refresh
    if args.atomic_rank_u != 0:
        tune["atomic_rank_u"] = args.atomic_rank_u
    if args.fresh_rank_u != 0:
        tune["fresh_rank_u"] = args.fresh_rank_u
    if args.refresh_rows != 0:
        tune["refresh_rows_u"] = args.refresh_rows

    def make_trainer(relaxed, loss=bool(args.loss), bitlevel=args.bitlevel):
        tr = w2b.Trainer(V, D, W, K, bitlevel, num_threads=nw_local,
                         iter=1, alpha=0.05, sample=0.0, reg=0.0, train_words=train_words * world,
                         compute_loss=loss, device=local_rank, worker_offset=worker_offset,
                         total_threads=nw_local * world, relaxed_coherence=relaxed,
                         window_cache=wcache, row_groups=rgroups, **tune)
        tr.init_net()                                  # InitNet values (LCG seed 1), ref :343-361
        tr.set_vocab_counts(counts, 100_000_000)       # 1e8-entry unigram table, ref :112-128
        return tr

    t = make_trainer(bool(args.relaxed))
    tuning_used = t.get_tuning()
    sync_impl = "none (1 GPU)"
    torch_sync = None
    if world > 1:
        # replicas: delta-sum exchange of [u||v] over RCCL, either through the process group that
        # torch.distributed.run set up (default) or through the library's own communicator
        ok = torch.ones(1, device=dev)
        try:
            if args.sync_impl != "lib" or shared:
                raise RuntimeError("torch sync requested")
            t.comm_init(world, rank, replicas.exchange_unique_id(dist, rank, w2b.comm_unique_id))
        except Exception as e:
            if args.sync_impl == "lib" and not shared:
                print("rank %d: library RCCL init failed (%r), using torch.distributed" % (rank, e), file=sys.stderr)
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() == 0:
            t.exchange_init()                                   # the replicas are identical here (InitNet)
            torch_sync = replicas.PhasedReplicaSync(dist, t, args.sync_mode)
            sync_impl = "library exchange kernels (w2b_exchange_*) + torch.distributed all_reduce (%s)%s" % (
                "RCCL" if backend == "nccl" else backend, " -- SMOKE TEST: ranks share GPUs, not a measurement" if shared else "")
        else:
            sync_impl = "library RCCL communicator (w2b_sync_replicas): chunked, on its own streams, overlapped with training"

    nsteps = args.steps + args.warmup
    kinfo = None
    kname = None
    if args.form == "tuples":
        B = args.batch
        need = nsteps * B
        reps = (need + per_rank_tokens - 1) // per_rank_tokens
        table_dev = None
        batches = []
        # unigram table on device for drawing negatives with the reference's table semantics
        import ctypes as C
        from word2bits_amd import _lib
        tab = np.empty(100_000_000, np.int32)
        _lib.check(_lib.lib().w2b_build_unigram_table(counts.ctypes.data_as(_lib.i64p), V,
                                                      tab.ctypes.data_as(_lib.i32p), len(tab)))
        table_dev = torch.from_numpy(tab).to(dev)
        del tab
        for s in range(nsteps):
            o = (s * B) % max(1, per_rank_tokens - B)
            center = stream[o:o + B].clone()
            center[center == 0] = 1
            ctx = draw_ids(torch, cdf, B * cw, gen)
            r = torch.randint(0, 100_000_000, (B * K,), device=dev, generator=gen)
            neg = table_dev[r]
            neg[neg == 0] = 1                                   # ref :457 remaps 0 to a random word
            cexp = center.repeat_interleave(K)
            neg = torch.where(neg == cexp, (neg % (V - 2)) + 1, neg)        # never the centre (ref :458)
            off = (torch.arange(B + 1, device=dev, dtype=torch.int32) * cw).contiguous()
            batches.append((center.contiguous(), off, ctx.contiguous(), neg.contiguous()))
        del table_dev
        words_per_step = B

        def prepare(tr):
            pass

        def step(i, tr=None):
            c, off, ctx, neg = batches[i]
            (tr or t).train_tuples_device(B, c.data_ptr(), off.data_ptr(), ctx.data_ptr(), neg.data_ptr(), 0.05,
                                          args.grid)
    else:
        def prepare(tr):
            tr.set_corpus_device(stream.data_ptr(), per_rank_tokens)
            # every rank holds its own stream: local shards are cut inside it
            tr.set_shards(replicas.token_shard_starts(per_rank_tokens, workers, 0, workers))
            tr.epoch_begin()

        prepare(t)
        kinfo = t.worker_kernel_info()
        kname = t.worker_kernel_name()
        positions = args.positions if args.positions > 0 else max(1, args.batch // workers)
        words_per_step = workers * positions

        def step(i, tr=None):
            # a shard lasts steps_per_epoch steps; after that the workers start their next epoch
            # (pthread_create of the next iteration, ref :532-535) -- any --steps value is valid
            steps_per_epoch = max(1, (per_rank_tokens // workers - 1100) // positions)
            if i > 0 and i % steps_per_epoch == 0:
                (tr or t).epoch_begin()
            (tr or t).train_step(positions)

    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    n_syncs = [0]

    def exchange():
        if torch_sync is not None:
            torch_sync.sync()
        else:
            t.sync_replicas(args.sync_mode)
        n_syncs[0] += 1

    def run(n0, n1, timed):
        # replicas exchange every --sync-every timed steps AND after the last timed step (the CLI also exchanges at
        # every epoch end), so the timed region always contains the exchange, whatever --steps is
        for i in range(n0, n1):
            step(i)
            if world > 1 and timed:
                done = i + 1 - args.warmup
                if done % args.sync_every == 0 or i + 1 == n1:
                    exchange()

    run(0, args.warmup, False)
    if world > 1:
        # one untimed exchange: first-use costs of the communicator stay out of the timed region, and a library
        # communicator that cannot exchange (all ranks must agree) is replaced by the torch.distributed path
        ok = torch.ones(1, device=dev)
        try:
            exchange()
        except Exception as e:
            if torch_sync is not None:
                raise
            print("rank %d: library exchange failed (%r), using torch.distributed" % (rank, e), file=sys.stderr)
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() == 0 and torch_sync is None:
            raise SystemExit("library exchange failed after its communicator was created")
        t.synchronize()
        n_syncs[0] = 0
        t.sync_stats()
    t.synchronize()
    t.timing_enable(True)
    t.timing_read()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.warmup, nsteps, True)
    t.synchronize()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    per_launch = np.sort(t.timing_launches())
    kernel_ms, launches = t.timing_read()
    sync_n, sync_ms = t.sync_stats() if world > 1 else (0, 0.0)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ranks of the collective that actually ran: RCCL's own count for the library communicator; the process group's
    # size when torch.distributed carried the sum over RCCL; 0 when no RCCL collective was involved (1 GPU, gloo hook)
    if world > 1 and torch_sync is None:
        rccl_ranks = t.comm_count()
    elif world > 1 and backend == "nccl":
        rccl_ranks = dist.get_world_size()
    else:
        rccl_ranks = 0
    total_words = words_per_step * args.steps * world
    value = total_words / dt
    bpw = algorithmic_bytes_per_word(D, cw, K)
    avg_launch_s = (kernel_ms / 1e3) / max(1, launches)
    achieved = words_per_step * bpw / avg_launch_s / 1e9        # GB/s, algorithmic
    result = {
        "metric": "training words/sec at dim=%d bitlevel=%d neg=%d" % (D, args.bitlevel, K),
        "value": value, "unit": "words/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "rccl_ranks": rccl_ranks,
        "config": {"workload": "%s: synthetic %dM-token %s stream, vocab=%d, bitlevel=%d, "
                               "size=%d, window=%d, negative=%d, sample=0; form=%s, %d centre words/step/GPU"
                               % (workload_name(args), args.tokens // 1_000_000, args.ids, V, args.bitlevel, D, W, K,
                                  args.form, words_per_step),
                   "form": args.form, "vocab": V, "dim": D, "window": W, "negative": K,
                   "bitlevel": args.bitlevel, "words_per_step_per_gpu": words_per_step, "ids": args.ids,
                   "row_coherence": "relaxed (plain cached accesses)" if args.relaxed else
                                    "agent scope (sc1): Hogwild coherent across the 8 XCD L2s",
                   "replica_sync": ("%s every %d steps, mode %d (0 delta-sum, 1 average, 2 saturation + quantization cells)" %
                                    (sync_impl, args.sync_every, args.sync_mode)) if world > 1 else sync_impl,
                   "exchanges_in_timed_region": n_syncs[0],
                   "worker_kernel": (dict(zip(("sentence_resident", "radius", "column_bytes", "workers_per_cu",
                                               "hot_rows_with_xcd_copies"), kinfo), kernel=kname) if kinfo else None),
                   "tuning": tuning_used,
                   "workers": workers if args.form == "worker" else None},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved * 1e9 / HBM_PEAK, "traffic": None,
                     "kernel": "k_train_tuples" if args.form == "tuples" else
                               ("k_train_resident" if kinfo and kinfo[0] else ("k_train_groups" if kname == "groups" else "k_train_workers")),
                     "algorithmic_bytes_per_word": bpw, "avg_launch_ms": avg_launch_s * 1e3,
                     "launches": launches,
                     "launch_ms_min_median_max": ([float(per_launch[0]), float(np.median(per_launch)), float(per_launch[-1])]
                                                  if len(per_launch) else None),
                     "loss_bookkeeping": bool(args.loss)},
    }
    if args.form == "worker":
        # one worker's pace beside one thread of the reference (cpu_baseline_configs0 / cpu_baseline_1thread)
        result["us_per_word_per_worker"] = 1e6 * workers / (words_per_step / avg_launch_s)
    result["roofline"]["algorithmic_bytes_per_launch"] = words_per_step * bpw
    # HBM bytes per launch from the counters: collected by tools/gpu_profile_session.sh with rocprofv3 --pmc (separate
    # FETCH_SIZE / WRITE_SIZE passes of this same command) and committed; quoted only for the shape they were measured on
    pmc = latest_profile("pmc_%s.json" % args.form)
    if world == 1 and os.path.exists(pmc) and args.ids == "zipf" and not args.relaxed:
        try:
            pj = json.load(open(pmc))
            same = (pj.get("vocab"), pj.get("dim"), pj.get("negative"), pj.get("bitlevel")) == (V, D, K, args.bitlevel) and \
                pj.get("kernel") == result["roofline"]["kernel"]
            if same and pj.get("words_per_launch"):
                per_word = pj["hbm_bytes_per_launch"] / pj["words_per_launch"]
                result["roofline"]["traffic"] = per_word * words_per_step
                result["roofline"]["traffic_GBps"] = per_word * words_per_step / avg_launch_s / 1e9
                result["roofline"]["traffic_frac"] = per_word * words_per_step / avg_launch_s / HBM_PEAK
                result["roofline"]["traffic_measured_in_this_run"] = False
                result["roofline"]["traffic_source"] = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate "
                                                        "passes; factors calibrated on known row bytes, profiles/r04_pmc_calibration.json: "
                                                        "FETCH x2.00, WRITE x1.00 for sc1 rows) on this command, %d centre "
                                                        "words per launch there" %
                                                        (os.path.basename(pmc), pj["words_per_launch"]))
        except Exception:
            pass
    if world > 1:
        words_between = words_per_step * args.sync_every
        xbytes = 8 * V * D
        result["replica_exchange"] = {
            "every_steps": args.sync_every, "centre_words_per_replica_between_exchanges": words_between,
            "mode": ["delta-sum", "average", "mode 2: exponential saturation per row decides every element's quantized value, the whole "
                     "sum where it stays in that quantization cell (round 6, DESIGN.md 3.5)"][args.sync_mode],
            "implementation": sync_impl, "exchanges": n_syncs[0],
            "bytes": xbytes, "bytes_all_reduced_per_exchange": xbytes,
            # cost model of one full exchange over xGMI beside the launches of one interval: a ring all-reduce is bound
            # by one ~153 GB/s link (2 (R-1)/R S bytes), a direct reduce-scatter + all-gather uses all R-1 links (2 S / R per link)
            "link_model_ms": {"ring_one_link": 2 * (world - 1) / world * xbytes / 153e9 * 1e3,
                              "direct_all_links": 2 * xbytes / world / 153e9 * 1e3,
                              "interval_ms": dt / args.steps * 1e3 * args.sync_every},
            # measured on one GPU through the phase API at this shape (8 replicas x 128 workers, tests/experiments/replica_rules.py,
            # tests/test_gpu_exchange.py; profiles/r06_sessions/): epoch loss against the single replica with the same 1024 workers
            "fidelity_measured_on_one_gpu": "8 replicas x 128 workers vs 1 replica x 1024: -0.5 % on the literal configs[1] stream at 1 M words "
                                            "per replica between exchanges, -2.9 % on the 22 M-token proxy at 131 K (round 5's rule: -12.6 % / -9.0 %)",
            "elementwise_cost_measured_on_one_gpu": "k_xchg_delta 2.2-7.2 ms + k_xchg_apply 2.9-3.3 ms per full exchange of 2.56 GB = 3-5 % of a 1 M-word launch at 128 workers (profiles/r05_sessions/r05m_replicas8.json)",
            "device_ms": (sync_ms / sync_n) if sync_n else None,
            "device_ms_per_exchange": (sync_ms / sync_n) if sync_n else None,
            # the library's own communicator runs the exchange on its own streams next to the training launches; with a
            # host-driven collective (torch.distributed) the host waits for every chunk
            "overlapped": torch_sync is None,
            "chunk_bytes": min(8 * V * D, 256 << 20)}
    t.close()

    def timed_leg(tr, wps):
        prepare(tr)
        for i in range(args.warmup):
            step(i, tr)
        tr.synchronize()
        tr.timing_enable(True)
        tr.timing_read()
        r0 = time.perf_counter()
        for i in range(args.warmup, nsteps):
            step(i, tr)
        tr.synchronize()
        rdt = time.perf_counter() - r0
        rms, rl = tr.timing_read()
        tr.close()
        return {"value": wps * args.steps / rdt, "unit": "words/s", "ms_per_step": rdt / args.steps * 1e3,
                "roofline_frac": wps * bpw / ((rms / 1e3) / max(1, rl)) / HBM_PEAK}

    ref1 = cpu_path = None
    if rank == 0 and world == 1 and args.cpu_baseline == "reference" and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "word2bits_stock")):
        # the CPU legs use one bounded corpus; the ONE-thread run of the reference (BASELINE.md section 4 asks for -threads 1
        # and -threads $(nproc)) starts now and runs on one host core next to the GPU legs below; the all-threads run and
        # the configs[0] run follow when the GPU is done, alone on the host
        try:
            cpu_path = write_cpu_corpus(args)
            ref1 = RefProbe(cpu_path, ref_flags(args), 1, sample_seconds=args.cpu_1thread_seconds)
        except Exception:
            ref1 = None
    if world == 1 and args.also_legs and args.form == "worker" and not args.relaxed:
        # the other instantiation of the same kernel: ./word2bits always books the loss (it prints "Epoch Loss", ref :539),
        # so the headline does too; a host that does not want the number can switch it off (compute_loss = 0)
        leg = timed_leg(make_trainer(False, loss=not args.loss), words_per_step)
        leg["note"] = "same steps with compute_loss = %d" % (not args.loss)
        result["without_loss_bookkeeping" if args.loss else "with_loss_bookkeeping"] = leg
        if args.bitlevel != 2:
            leg = timed_leg(make_trainer(False, bitlevel=2), words_per_step)
            leg["note"] = "same steps at bitlevel 2 (BASELINE configs[2] quantizer)"
            result["bitlevel2"] = leg
    if world == 1 and args.also_relaxed and not args.relaxed:
        # same steps, same data, relaxed row coherence -- reported beside the headline
        if args.form == "worker":          # relaxed rows run the plain worker kernel with its own residency
            probe = w2b.Trainer(2, D, W, K, args.bitlevel, num_threads=1, device=local_rank, relaxed_coherence=True,
                                window_cache=wcache, compute_loss=False)
            nw_local = workers = probe.suggested_threads() if args.workers <= 0 else workers
            probe.close()
            positions = args.positions if args.positions > 0 else max(1, args.batch // workers)
            words_per_step = workers * positions
        t2 = make_trainer(True)
        prepare(t2)
        for i in range(args.warmup):
            step(i, t2)
        t2.synchronize()
        t2.timing_enable(True)
        t2.timing_read()
        r0 = time.perf_counter()
        for i in range(args.warmup, nsteps):
            step(i, t2)
        t2.synchronize()
        rdt = time.perf_counter() - r0
        rms, rl = t2.timing_read()
        result["relaxed_coherence"] = {
            "value": words_per_step * args.steps / rdt, "unit": "words/s", "ms_per_step": rdt / args.steps * 1e3,
            "roofline_frac": words_per_step * bpw / ((rms / 1e3) / max(1, rl)) / HBM_PEAK,
            "note": "plain cached row accesses: hot rows are private per XCD L2 within a launch (not the default)"}
        t2.close()
    if (world == 1 and args.also_shapes and workload_name(args) == "BASELINE configs[1]" and args.form == "worker"
            and not args.relaxed and args.loss and not tune):
        result["other_shapes"] = other_shapes()
    if rank == 0:
        cb = None
        if world == 1 and args.cpu_baseline != "none":
            try:
                if ref1 is not None:
                    r1 = ref1.result()
                    result["cpu_baseline_1thread"] = None if not r1 else {
                        "value": r1["words_per_s"], "unit": "words/s", "cores": 1, "kind": "reference",
                        "sample": "the same program and corpus with -threads 1: %d words in %.1f s after 'Starting epoch' (progress "
                                  "line, ref :384-387), then stopped; it ran on one host core beside the GPU legs" % (r1["words"], r1["train_s"])}
                if args.cpu_baseline == "reference" and cpu_path:
                    cb = cpu_baseline_reference(args, cpu_path)
                if cb is None:
                    cb = cpu_baseline_port(args)
                if args.cpu_baseline == "reference" and args.cpu_cfg0:
                    result["cpu_baseline_configs0"] = cpu_baseline_cfg0()
            except Exception as e:            # the GPU number must still be reported
                cb = {"value": None, "unit": "words/s", "cores": os.cpu_count(), "kind": "port",
                      "sample": "failed: %r" % (e,)}
            finally:
                if cpu_path:
                    try:
                        os.remove(cpu_path)
                        os.rmdir(os.path.dirname(cpu_path))
                    except OSError:
                        pass
        result["cpu_baseline"] = cb
        # end-to-end wall times of ./word2bits and the reference program on one file (tools/e2e_compare.py): the GPU side
        # (3 s) is measured in THIS run; the reference side (167 s on 256 host threads) is quoted from the committed file
        if world == 1 and workload_name(args).startswith("BASELINE configs[1]") and args.also_shapes:
            try:
                quoted = json.load(open(latest_profile("e2e.json")))
                tmp = os.path.join("/tmp", "w2b_e2e_%d.json" % os.getpid())
                subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_compare.py"), tmp, "--only", "hip"],
                               capture_output=True, timeout=300)
                mine = json.load(open(tmp))
                os.remove(tmp)
                result["e2e"] = {"file": mine.get("file"), "flags": mine.get("flags"), "word2bits_hip": mine.get("word2bits_hip"),
                                 "word2bits_hip_measured_in_this_run": True,
                                 "reference": quoted.get("reference"), "reference_threads": quoted.get("reference_threads"),
                                 "reference_source": "quoted from " + os.path.relpath(latest_profile("e2e.json"), ROOT)}
                if mine.get("word2bits_hip") and quoted.get("reference"):
                    result["e2e"]["speedup_total"] = round(quoted["reference"]["total_s"] / mine["word2bits_hip"]["total_s"], 1)
                    result["e2e"]["speedup_train"] = round(quoted["reference"]["train_s"] / max(mine["word2bits_hip"]["train_s"], 1e-3), 1)
            except Exception as e:
                result["e2e"] = {"error": repr(e)[:200]}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
