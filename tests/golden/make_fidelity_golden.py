"""Generates tests/golden/fidelity_golden.json: what the UNMODIFIED reference program (oracle/_ref/word2bits_stock,
built from /root/reference by oracle/Makefile) does on the planted-analogy corpus (tests/planted.py) under Hogwild
with several thread counts, scored by the unmodified evaluator (oracle/_ref/compute_accuracy).  The GPU tests
(tests/test_gpu_fidelity.py) hold the HIP trainer to these bands: per-epoch losses and total accuracy of runs with the
same number of workers.  Multi-threaded runs are racy, so every configuration is run several times and the band is
[min, max] over the runs.

Run here (container with /root/reference): python tests/golden/make_fidelity_golden.py
"""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from planted import make_planted, parse_accuracy

REF = os.path.join(ROOT, "oracle", "_ref")
TMP = "/tmp/w2b_fidelity_golden"
os.makedirs(TMP, exist_ok=True)
corpus, questions = os.path.join(TMP, "planted.txt"), os.path.join(TMP, "questions.txt")
ntok = make_planted(corpus, questions, repeats=120)

CONFIGS = [
    # name, flags, thread counts, runs per thread count
    ("b1_d200", dict(bitlevel=1, size=200, window=8, negative=24, iter=5), [1, 8, 64, 512], 3),
    # BASELINE configs[2] shape: bitlevel 2, size 400, negative 24, iter 5 (text8 itself is not available offline)
    ("cfg2_b2_d400", dict(bitlevel=2, size=400, window=8, negative=24, iter=5), [8, 64], 3),
]
out = {"corpus": {"generator": "tests/planted.py make_planted(repeats=120, seed=0)", "tokens": ntok},
       "program": "oracle/_ref/word2bits_stock (unmodified reference, -O3 -march=x86-64-v3)", "configs": {}}
for name, fl, threads, runs in CONFIGS:
    flags = []
    for k, v in fl.items():
        flags += ["-" + k, str(v)]
    flags += ["-min-count", "5", "-binary", "1"]
    res = {"flags": fl, "runs": []}
    for th in threads:
        for r in range(runs if th > 1 else 1):
            vec = os.path.join(TMP, "ref.bin")
            t0 = time.time()
            p = subprocess.run([os.path.join(REF, "word2bits_stock"), "-train", corpus, "-output", vec,
                                "-threads", str(th)] + flags, capture_output=True, text=True, check=True)
            losses = [float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", p.stdout)]
            with open(questions) as q:
                acc = parse_accuracy(subprocess.run([os.path.join(REF, "compute_accuracy"), vec, "0", "0"], stdin=q,
                                                    capture_output=True, text=True).stdout)
            rec = {"threads": th, "epoch_losses": losses, "accuracy": acc, "secs": round(time.time() - t0, 1)}
            print(name, json.dumps(rec), flush=True)
            res["runs"].append(rec)
    out["configs"][name] = res
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "fidelity_golden.json"), "w"), indent=1)
