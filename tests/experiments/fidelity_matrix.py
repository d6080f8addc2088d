"""Measurement, not a test: epoch losses of ./word2bits against the reference bands of tests/golden/fidelity_bands.json
(recorded on the GPU box's 256-thread host) over worker counts x kernels x the lossless-update knobs.
usage: python tests/experiments/fidelity_matrix.py [text8size] [headline] [planted]   (run on the GPU box)"""
import json, os, re, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from w2b_testlib import write_zipf_text_corpus, write_headline_corpus
from planted import make_planted
B = json.load(open(os.path.join(ROOT, "tests", "golden", "fidelity_bands.json")))["jobs"]
CLI = os.path.join(ROOT, "word2bits")
TMP = "/tmp/w2b_fid"
os.makedirs(TMP, exist_ok=True)


def band(job, threads):
    L = np.array([r["epoch_losses"] for r in B[job]["runs"] if r["threads"] == threads])
    return L.mean(0), L.std(0, ddof=1) if len(L) > 1 else np.zeros(L.shape[1])


def run(corpus, flags, threads, extra):
    args = [CLI, "-train", corpus, "-output", os.path.join(TMP, "o.bin"), "-threads", str(threads), "-min-count", "5", "-binary", "1"]
    for k, v in flags.items():
        args += ["-" + k, str(v)]
    t0 = time.time()
    r = subprocess.run(args + extra, capture_output=True, text=True)
    if r.returncode != 0:
        return None, r.stderr[-200:]
    w = re.search(r"Hogwild workers \(workgroups\): (\d+)", r.stdout)
    return np.array([float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", r.stdout)]), (int(w.group(1)) if w else threads, time.time() - t0)


def matrix(job, corpus, flags, cases, refs):
    for th in refs:
        m, s = band(job, th)
        print("%-58s %s  (rel. std %s %%)" % ("reference %d threads" % th, np.round(m / 1e3).tolist(), np.round(100 * s / np.abs(m), 2).tolist()))
    for name, th, extra, ref_th in cases:
        losses, info = run(corpus, flags, th, extra)
        if losses is None:
            print("%-58s FAILED %s" % (name, info)); continue
        m, _ = band(job, ref_th)
        print("%-58s %s  vs ref@%d: %s %%   [%d workers, %.1f s]" % (name, np.round(losses / 1e3).tolist(), ref_th,
              np.round(100 * (losses - m) / np.abs(m), 2).tolist(), info[0], info[1]), flush=True)


jobs = [a for a in sys.argv[1:] if not a.startswith("-")] or ["text8size", "headline"]
EXTRA = [a for a in sys.argv[1:] if a.startswith("-")]          # e.g. -hot-period=32  ->  "-hot-period 32" on every run
EXTRA = sum(([a.split("=")[0], a.split("=")[1]] for a in EXTRA), [])
K = {"resident": ["-window-cache", "1"], "plain": ["-window-cache", "0"], "auto": []}
if "text8size" in jobs:
    c = write_zipf_text_corpus(os.path.join(TMP, "t8.txt"))
    fl = dict(bitlevel=1, size=200, window=8, negative=24, iter=3)
    cases = [("text8size threads=%d %s %s" % (th, k, " ".join(EXTRA)), th, K[k] + EXTRA, ref)
             for th, ref in ((0, 256), (256, 256), (64, 64)) for k in (("auto",) if th == 0 else ("resident", "plain"))]
    matrix("text8size", c, fl, cases, (64, 256))
    os.remove(c)
if "headline" in jobs:
    c = write_headline_corpus(os.path.join(TMP, "hl.txt"))
    fl = dict(bitlevel=1, size=800, window=8, negative=24, iter=1, sample=0)
    counts = [int(x) for x in os.environ.get("W2B_MATRIX_THREADS", "0,256,64").split(",")]      # 0 = -threads 0
    cases = [("headline threads=%d auto %s" % (th, " ".join(EXTRA)), th, EXTRA, 64 if 0 < th <= 64 else 256) for th in counts]
    matrix("headline", c, fl, cases, (64, 256))
    os.remove(c)
