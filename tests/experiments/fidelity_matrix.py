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


jobs = sys.argv[1:] or ["text8size", "headline"]
K = {"resident": ["-window-cache", "1"], "plain": ["-window-cache", "0"], "auto": []}
if "text8size" in jobs:
    c = write_zipf_text_corpus(os.path.join(TMP, "t8.txt"))
    fl = dict(bitlevel=1, size=200, window=8, negative=24, iter=3)
    cases = [("text8size threads=0 auto", 0, [], 256),
             ("text8size threads=0 plain hot-weight=500", 0, K["plain"] + ["-hot-weight", "500"], 256),
             ("text8size threads=0 plain hot-weight=1000", 0, K["plain"] + ["-hot-weight", "1000"], 256)]
    matrix("text8size", c, fl, cases, (256,))
    os.remove(c)
if "headline" in jobs:
    c = write_headline_corpus(os.path.join(TMP, "hl.txt"))
    fl = dict(bitlevel=1, size=800, window=8, negative=24, iter=1, sample=0)
    cases = [("headline threads=0 auto", 0, [], 256)]
    for th, ref in ((512, 256), (256, 256), (64, 64)):
        cases.append(("headline threads=%d auto" % th, th, [], ref))
    for name, extra in (("hot-weight=250", ["-hot-weight", "250"]), ("hot-weight=500", ["-hot-weight", "500"]),
                        ("hot-weight=1000", ["-hot-weight", "1000"]), ("hot-period=2", ["-hot-period", "2"]),
                        ("hot-period=32", ["-hot-period", "32"]), ("hot-cap=16", ["-hot-cap", "16"]), ("hot-cap=128", ["-hot-cap", "128"])):
        cases.append(("headline threads=0 plain %s" % name, 0, K["plain"] + extra, 256))
    for name, extra in (("hot-weight=500", ["-hot-weight", "500"]), ("hot-weight=1000", ["-hot-weight", "1000"])):
        cases.append(("headline threads=256 plain %s" % name, 256, K["plain"] + extra, 256))
        cases.append(("headline threads=64 plain %s" % name, 64, K["plain"] + extra, 64))
    matrix("headline", c, fl, cases, (64, 256))
    os.remove(c)
