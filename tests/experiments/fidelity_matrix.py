"""Measurement, not a test: epoch losses of ./word2bits against the reference bands (tests/golden/fidelity_bands.json,
recorded on the GPU box's 256-thread host; --bands adds a file recorded in the same session) over
regimes x worker counts x ARMS, an arm being a set of extra command-line flags (the w2b_tuning knobs).

usage (on the GPU box):
  python tests/experiments/fidelity_matrix.py --jobs headline,heldout_k5 --threads 0,256,64 \
         --arms "default:;lossless:-atomic-rank 300 -atomic-rank-u 1000" [--bands gpurun_out/bands.json] [--kernel plain]
Prints one line per (job, threads, arm): losses, deviation from the reference's mean at the matching thread count
(64 -> the 64-thread band, else the 256-thread band), workers, seconds; and a JSON record per line into --out."""
import argparse, json, os, re, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from w2b_testlib import write_zipf_text_corpus, write_headline_corpus, write_heldout_corpus, HELDOUT
CLI = os.path.join(ROOT, "word2bits")
TMP = "/tmp/w2b_fid"
os.makedirs(TMP, exist_ok=True)

ap = argparse.ArgumentParser()
ap.add_argument("--jobs", default="headline")
ap.add_argument("--threads", default="0,256,64")
ap.add_argument("--arms", default="default:")
ap.add_argument("--bands", default="")
ap.add_argument("--kernel", default="auto", help="auto | plain | resident | both")
ap.add_argument("--out", default="")
a = ap.parse_args()
B = json.load(open(os.path.join(ROOT, "tests", "golden", "fidelity_bands.json")))["jobs"]
if a.bands and os.path.exists(a.bands):
    B.update(json.load(open(a.bands))["jobs"])
ARMS = []
for part in a.arms.split(";"):
    if part.strip():
        name, _, fl = part.partition(":")
        ARMS.append((name.strip(), fl.split()))
K = {"resident": ["-window-cache", "1"], "plain": ["-window-cache", "0"], "auto": []}
KERNELS = ["plain", "resident"] if a.kernel == "both" else [a.kernel]


def band(job, threads):
    if job not in B:
        return None, None
    L = np.array([r["epoch_losses"] for r in B[job]["runs"] if r["threads"] == threads])
    if len(L) == 0:
        return None, None
    return L.mean(0), (L.std(0, ddof=1) if len(L) > 1 else np.zeros(L.shape[1]))


def run(corpus, flags, threads, extra):
    args = [CLI, "-train", corpus, "-output", "/dev/null", "-threads", str(threads), "-min-count", "5", "-binary", "1"]
    for k, v in flags.items():
        args += ["-" + k, str(v)]
    t0 = time.time()
    r = subprocess.run(args + extra, capture_output=True, text=True)
    if r.returncode != 0:
        return None, (r.stdout[-100:] + r.stderr[-200:])
    w = re.search(r"Hogwild workers \(workgroups\): (\d+)", r.stdout)
    return np.array([float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", r.stdout)]), (int(w.group(1)) if w else threads, time.time() - t0)


def corpus_of(job):
    if job == "headline":
        return write_headline_corpus(os.path.join(TMP, "hl.txt")), dict(bitlevel=1, size=800, window=8, negative=24, iter=1, sample=0)
    if job == "text8size":
        return write_zipf_text_corpus(os.path.join(TMP, "t8.txt")), dict(bitlevel=1, size=200, window=8, negative=24, iter=3)
    if job.startswith("planted"):
        from planted import make_planted
        c = os.path.join(TMP, "planted.txt")
        make_planted(c, os.path.join(TMP, "planted_q.txt"), repeats=120)
        return c, B[job]["flags"]
    return write_heldout_corpus(os.path.join(TMP, job + ".txt"), job), HELDOUT[job]["flags"]


out = open(a.out, "a") if a.out else None
for job in a.jobs.split(","):
    c, fl = corpus_of(job)
    for th in (8, 64, 256):
        m, s = band(job, th)
        if m is not None:
            print("%-14s reference %3d threads %s  (rel. std %s %%)" % (job, th, np.round(m / 1e3).tolist(), np.round(100 * s / np.abs(m), 2).tolist()))
    for th in [int(x) for x in a.threads.split(",")]:
        ref_th = (8 if 0 < th <= 8 else 64) if 0 < th <= 64 else 256
        m, _ = band(job, ref_th)
        for kern in KERNELS:
            for name, extra in ARMS:
                losses, info = run(c, fl, th, K[kern] + extra)
                if losses is None:
                    print("%-14s threads=%-4d %-8s %-28s FAILED %s" % (job, th, kern, name, info), flush=True)
                    continue
                dev = (100 * (losses - m) / np.abs(m)) if m is not None and len(m) == len(losses) else None
                print("%-14s threads=%-4d %-8s %-28s %s  vs ref@%d: %s %%   [%d workers, %.1f s]" % (
                    job, th, kern, name, np.round(losses / 1e3).tolist(), ref_th,
                    np.round(dev, 2).tolist() if dev is not None else "n/a", info[0], info[1]), flush=True)
                if out:
                    out.write(json.dumps({"job": job, "threads": th, "kernel": kern, "arm": name, "flags": extra, "losses": losses.tolist(),
                                          "dev_pct": dev.tolist() if dev is not None else None, "workers": info[0], "secs": info[1]}) + "\n")
                    out.flush()
    os.remove(c)
