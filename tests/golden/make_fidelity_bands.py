"""Reference bands for the Hogwild fidelity tests (tests/test_gpu_fidelity.py): what the UNMODIFIED reference program
(oracle/_ref/word2bits_stock, built from /root/reference by oracle/Makefile; the binary travels to the GPU box) does
with N truly concurrent threads.

The round-2 bands were recorded in the 8-vCPU build container, where 64 or 512 threads are
time-sliced: each thread runs alone for milliseconds, which is far less concurrent than 64 workgroups of a GPU.  This
script is meant to run on the GPU box's HOST (2 x EPYC 9575F, 256 hardware threads; `gpurun -- python
tests/golden/make_fidelity_bands.py --out gpurun_out/bands.json ...`), so that "the same thread count" also means the
same concurrency.  It needs no GPU and no /root/reference; every run is repeated so that the tests can derive their
tolerances from the reference's own run-to-run spread.

jobs (any subset, --jobs a,b,c):
  headline    BASELINE configs[1] shape: V=400 K, D=800, window 8, negative 24, bitlevel 1, -sample 0, -iter 1 on
              w2b_testlib.write_headline_corpus (every word 5x + a Zipf(1) stream)
  text8size   17 M Zipf tokens over 70 K words, bitlevel 1, size 200, negative 24, -iter 3
  planted     planted-analogy corpus, bitlevel 1 size 200 and bitlevel 2 size 400, -iter 5, scored by the unmodified
              evaluator
  heldout_k5, heldout_zipf12   (round 4) the held-out regimes of w2b_testlib.HELDOUT: V=100 K, size 300, window 5,
              negative 5, -sample 0; and Zipf exponent 1.2 at the configs[2] shape (bitlevel 2, size 400, negative 24)
Each job takes "threads x runs" pairs: --headline 64x3,256x2 ...; runs of one thread count that fit side by side on the
host's hardware threads (threads * runs <= cpu_count) are started concurrently.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from planted import make_planted, parse_accuracy                                      # noqa: E402
from w2b_testlib import write_headline_corpus, write_zipf_text_corpus, write_heldout_corpus, HELDOUT, HELDOUT_BIG   # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def spec(s):
    out = []
    for part in s.split(","):
        if part:
            th, runs = part.split("x")
            out.append((int(th), int(runs)))
    return out


def run_many(corpus, flags, threads, runs, tmp, questions=None, serial=False):
    """`runs` runs at `threads` threads; concurrently when they fit on the host"""
    ncpu = os.cpu_count() or 1
    batch = 1 if serial else max(1, min(runs, ncpu // max(1, threads)))
    recs = []
    for r0 in range(0, runs, batch):
        procs = []
        for r in range(r0, min(runs, r0 + batch)):
            vec = os.path.join(tmp, "ref_%d_%d.bin" % (threads, r)) if questions else "/dev/null"
            cmd = [os.path.join(REF, "word2bits_stock"), "-train", corpus, "-output", vec, "-threads", str(threads)] + flags
            procs.append((time.time(), vec, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)))
        for t0, vec, p in procs:
            out = p.communicate()[0]
            rec = {"threads": threads, "concurrent_runs": len(procs),
                   "epoch_losses": [float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", out)],
                   "secs": round(time.time() - t0, 1)}
            m = re.search(r"Vocab size: (\d+)", out)
            rec["vocab_size"] = int(m.group(1)) if m else None
            m = re.search(r"Words in train file: (\d+)", out)
            rec["train_words"] = int(m.group(1)) if m else None
            if questions:
                with open(questions) as q:
                    rec["accuracy"] = parse_accuracy(subprocess.run([os.path.join(REF, "compute_accuracy"), vec, "0", "0"],
                                                                    stdin=q, capture_output=True, text=True).stdout)
                os.remove(vec)
            print(json.dumps(rec), flush=True)
            recs.append(rec)
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--jobs", default="headline,text8size,planted")
    ap.add_argument("--headline", default="64x3,256x2")
    ap.add_argument("--headline-zipf-tokens", type=int, default=20_000_000)
    ap.add_argument("--text8size", default="64x2,256x2")
    ap.add_argument("--planted", default="8x3,64x3,512x3")
    ap.add_argument("--heldout", default="64x2,256x2", help="threads x runs for the held-out regimes (jobs heldout_k5, heldout_zipf12)")
    ap.add_argument("--cfg1", default="256x1", help="threads x runs for job cfg1_100m: BASELINE configs[1] LITERALLY -- the 100 M-token stream "
                                                    "bench.py times (every word 5x + 98 M Zipf(1) tokens), ~13 minutes of a 256-thread host per run")
    ap.add_argument("--heldout-big", default="256x1", help="threads x runs for job heldout_k5_big (60 M tokens, ~5.5 minutes per run)")
    ap.add_argument("--reuse-corpus", default="", help="HELDOUT_BIG jobs: an existing corpus file (written by write_heldout_corpus for one of "
                                                       "the jobs named) is used as it is and left in place -- a session that runs ./word2bits on the same file "
                                                       "writes it once")
    ap.add_argument("--tmp", default="/tmp/w2b_bands")
    a = ap.parse_args()
    os.makedirs(a.tmp, exist_ok=True)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    host = {"cpus": os.cpu_count()}
    try:
        host["model"] = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    res = {"program": "oracle/_ref/word2bits_stock (unmodified reference, -O3 -march=x86-64-v3)", "host": host, "jobs": {}}

    def flush():
        json.dump(res, open(a.out, "w"), indent=1)

    jobs = a.jobs.split(",")
    if "planted" in jobs:
        corpus, questions = os.path.join(a.tmp, "planted.txt"), os.path.join(a.tmp, "questions.txt")
        ntok = make_planted(corpus, questions, repeats=120)
        for name, fl in (("b1_d200", dict(bitlevel=1, size=200, window=8, negative=24, iter=5)),
                         ("cfg2_b2_d400", dict(bitlevel=2, size=400, window=8, negative=24, iter=5))):
            flags = sum((["-" + k, str(v)] for k, v in fl.items()), []) + ["-min-count", "5", "-binary", "1"]
            job = {"corpus": "tests/planted.py make_planted(repeats=120, seed=0): %d tokens" % ntok, "flags": fl, "runs": []}
            for th, runs in spec(a.planted):
                # (serial: 564 K tokens take seconds, and runs side by side would share memory bandwidth unevenly)
                job["runs"] += run_many(corpus, flags, th, runs, a.tmp, questions, serial=True)
            res["jobs"]["planted_" + name] = job
            flush()
    if "text8size" in jobs:
        corpus = write_zipf_text_corpus(os.path.join(a.tmp, "text8size.txt"))
        fl = dict(bitlevel=1, size=200, window=8, negative=24, iter=3)
        flags = sum((["-" + k, str(v)] for k, v in fl.items()), []) + ["-min-count", "5", "-binary", "1"]
        job = {"corpus": "write_zipf_text_corpus(vocab=70000, n_tokens=17_000_000, seed=0)", "flags": fl, "runs": []}
        for th, runs in spec(a.text8size):
            job["runs"] += run_many(corpus, flags, th, runs, a.tmp)
            flush()
        res["jobs"]["text8size"] = job
        os.remove(corpus)
    for name in HELDOUT:                  # round 4: regimes no knob was ever swept on (w2b_testlib.HELDOUT)
        if name in jobs:
            corpus = write_heldout_corpus(os.path.join(a.tmp, name + ".txt"), name)
            fl = HELDOUT[name]["flags"]
            flags = sum((["-" + k, str(v)] for k, v in fl.items()), []) + ["-min-count", "5", "-binary", "1"]
            job = {"corpus": "write_heldout_corpus(%r): %r" % (name, HELDOUT[name]["corpus"]), "flags": fl, "runs": []}
            res["jobs"][name] = job
            for th, runs in spec(a.heldout):
                job["runs"] += run_many(corpus, flags, th, runs, a.tmp)
                flush()
            os.remove(corpus)
    if "headline" in jobs:
        corpus = write_headline_corpus(os.path.join(a.tmp, "headline.txt"), n_zipf=a.headline_zipf_tokens)
        fl = dict(bitlevel=1, size=800, window=8, negative=24, iter=1, sample=0)
        flags = sum((["-" + k, str(v)] for k, v in fl.items()), []) + ["-min-count", "5", "-binary", "1"]
        job = {"corpus": "write_headline_corpus(vocab=400000, n_zipf=%d, seed=1234)" % a.headline_zipf_tokens, "flags": fl, "runs": []}
        res["jobs"]["headline"] = job
        for th, runs in spec(a.headline):
            job["runs"] += run_many(corpus, flags, th, runs, a.tmp)
            flush()
        os.remove(corpus)
    if "cfg1_100m" in jobs:               # round 5: the benchmarked setting itself, not its 22 M-token proxy
        corpus = write_headline_corpus(os.path.join(a.tmp, "cfg1_100m.txt"), n_zipf=98_000_000)
        fl = dict(bitlevel=1, size=800, window=8, negative=24, iter=1, sample=0)
        flags = sum((["-" + k, str(v)] for k, v in fl.items()), []) + ["-min-count", "5", "-binary", "1"]
        job = {"corpus": "write_headline_corpus(vocab=400000, n_zipf=98000000, seed=1234): BASELINE configs[1] literally", "flags": fl, "runs": []}
        res["jobs"]["cfg1_100m"] = job
        for th, runs in spec(a.cfg1):
            job["runs"] += run_many(corpus, flags, th, runs, a.tmp, serial=True)
            flush()
        os.remove(corpus)
    for name in HELDOUT_BIG:              # round 4 recorded ONE run of this; round 5 adds to it
        if name in jobs:
            corpus = a.reuse_corpus or write_heldout_corpus(os.path.join(a.tmp, name + ".txt"), name)
            fl = HELDOUT_BIG[name]["flags"]
            flags = sum((["-" + k, str(v)] for k, v in fl.items()), []) + ["-min-count", "5", "-binary", "1"]
            job = {"corpus": "write_heldout_corpus(%r): %r" % (name, HELDOUT_BIG[name]["corpus"]), "flags": fl, "runs": []}
            res["jobs"][name] = job
            for th, runs in spec(a.heldout_big):
                job["runs"] += run_many(corpus, flags, th, runs, a.tmp, serial=True)
                flush()
            if not a.reuse_corpus:
                os.remove(corpus)
    flush()


if __name__ == "__main__":
    main()
