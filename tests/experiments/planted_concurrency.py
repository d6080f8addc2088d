"""Measurement, not a test: the planted corpus (small flat vocabulary: every row collides) at 64 / 512 workers with different
numbers of workers running AT ONCE (w2b_tuning.concurrent_workers / ./word2bits -concurrent N): epoch losses against the
unmodified reference's band at the same thread count (tests/golden/fidelity_bands.json) and the analogy accuracy.
  python tests/experiments/planted_concurrency.py [--out file.json]"""
import argparse, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from planted import make_planted, parse_accuracy

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="")
ap.add_argument("--cfg2-64", default="0,64,32,16,8", help="workers at once to try for planted_cfg2_b2_d400 at 64 threads (0 = automatic)")
a = ap.parse_args()
BANDS = json.load(open(os.path.join(ROOT, "tests", "golden", "fidelity_bands.json")))["jobs"]
d = tempfile.mkdtemp()
corpus, questions = os.path.join(d, "planted.txt"), os.path.join(d, "questions.txt")
make_planted(corpus, questions, repeats=120)
res = []
for job, threads, concs in (("planted_cfg2_b2_d400", 64, tuple(int(x) for x in a.cfg2_64.split(","))), ("planted_b1_d200", 64, (0, 64, 16)),
                            ("planted_b1_d200", 512, (0, 512, 64)), ("planted_cfg2_b2_d400", 8, (0,))):
    runs = [r for r in BANDS[job]["runs"] if r["threads"] == threads]
    mean = np.array([r["epoch_losses"] for r in runs]).mean(0)
    acc_ref = [r["accuracy"]["total"] for r in runs]
    for conc in concs:
        out = os.path.join(d, "o.bin")
        args = [os.path.join(ROOT, "word2bits"), "-train", corpus, "-output", out, "-threads", str(threads), "-min-count", "5", "-binary", "1"]
        for k, v in BANDS[job]["flags"].items():
            args += ["-" + k, str(v)]
        if conc:
            args += ["-concurrent", str(conc)]
        r = subprocess.run(args, capture_output=True, text=True)
        losses = np.array([float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", r.stdout)])
        with open(questions, "rb") as q:
            acc = parse_accuracy(subprocess.run([os.path.join(ROOT, "compute_accuracy"), out, "0", "0"], input=q.read(), capture_output=True).stdout.decode())
        dev = 100 * (losses - mean) / np.abs(mean)
        rec = {"job": job, "threads": threads, "concurrent": conc or "automatic", "deviation_pct": np.round(dev, 2).tolist(), "accuracy": acc["total"], "reference_accuracy": acc_ref}
        res.append(rec)
        print("PC %-22s threads %3d concurrent %-9s: deviation %s %%  accuracy %.2f (reference %s)" % (job, threads, conc or "automatic", np.round(dev, 2).tolist(), acc["total"], np.round(acc_ref, 1).tolist()), flush=True)
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
