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


jobs = sys.argv[1:] or ["text8size", "headline", "planted"]
K = {"resident": ["-window-cache", "1"], "plain": ["-window-cache", "0"]}
if "text8size" in jobs:
    c = write_zipf_text_corpus(os.path.join(TMP, "t8.txt"))
    fl = dict(bitlevel=1, size=200, window=8, negative=24, iter=3)
    cases = []
    for th, ref in ((0, 256), (256, 256), (64, 64)):
        for kern in ("resident", "plain"):
            cases.append(("text8size threads=%d %s (defaults)" % (th, kern), th, K[kern], ref))
    for name, extra in (("hot-rows=0", ["-hot-rows", "0"]), ("window-refresh=0", ["-window-refresh", "0"]),
                        ("hot-weight=500", ["-hot-weight", "500"]), ("hot-period=2", ["-hot-period", "2"]),
                        ("hot-rows=0 window-refresh=0", ["-hot-rows", "0", "-window-refresh", "0"])):
        cases.append(("text8size threads=0 resident %s" % name, 0, K["resident"] + extra, 256))
    cases.append(("text8size threads=0 plain hot-rows=0", 0, K["plain"] + ["-hot-rows", "0"], 256))
    matrix("text8size", c, fl, cases, (64, 256))
    os.remove(c)
if "headline" in jobs:
    c = write_headline_corpus(os.path.join(TMP, "hl.txt"))
    fl = dict(bitlevel=1, size=800, window=8, negative=24, iter=1, sample=0)
    cases = []
    for th, ref in ((0, 256), (256, 256), (64, 64)):
        for kern in ("resident", "plain"):
            cases.append(("headline threads=%d %s (defaults)" % (th, kern), th, K[kern], ref))
    for name, extra in (("hot-rows=0", ["-hot-rows", "0"]), ("window-refresh=0", ["-window-refresh", "0"]),
                        ("window-refresh=4", ["-window-refresh", "4"]),
                        ("hot-weight=500", ["-hot-weight", "500"]), ("hot-weight=1000", ["-hot-weight", "1000"]),
                        ("hot-period=2", ["-hot-period", "2"]),
                        ("hot-rows=0 window-refresh=0", ["-hot-rows", "0", "-window-refresh", "0"])):
        cases.append(("headline threads=0 resident %s" % name, 0, K["resident"] + extra, 256))
    cases.append(("headline threads=0 plain hot-rows=0", 0, K["plain"] + ["-hot-rows", "0"], 256))
    cases.append(("headline threads=64 resident hot-rows=0 window-refresh=0", 64, K["resident"] + ["-hot-rows", "0", "-window-refresh", "0"], 64))
    cases.append(("headline threads=64 plain hot-rows=0", 64, K["plain"] + ["-hot-rows", "0"], 64))
    matrix("headline", c, fl, cases, (64, 256))
    os.remove(c)
if "planted" in jobs:
    corpus, q = os.path.join(TMP, "pl.txt"), os.path.join(TMP, "q.txt")
    make_planted(corpus, q, repeats=120)
    for job, fl in (("planted_b1_d200", dict(bitlevel=1, size=200, window=8, negative=24, iter=5)),
                    ("planted_cfg2_b2_d400", dict(bitlevel=2, size=400, window=8, negative=24, iter=5))):
        cases = []
        for th in (8, 64, 512):
            for kern in ("resident", "plain"):
                cases.append(("%s threads=%d %s (defaults)" % (job, th, kern), th, K[kern], th))
        cases.append(("%s threads=64 resident atomic-rank=0" % job, 64, K["resident"] + ["-atomic-rank", "0"], 64))
        matrix(job, corpus, fl, cases, (8, 64, 512))
