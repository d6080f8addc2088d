"""Accuracy parity experiment (runs on the GPU box): the unmodified reference CPU program vs the HIP
trainer on the planted-analogy corpus, both scored by the UNMODIFIED reference evaluator
(oracle/_ref/compute_accuracy).  Prints one JSON line per run.

usage: python tests/experiments/accuracy_experiment.py [--bitlevel 1] [--size 200] [--iter 5] [--variants ,_sc1]
"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from planted import make_planted, parse_accuracy

ap = argparse.ArgumentParser()
ap.add_argument("--bitlevel", type=int, default=1)
ap.add_argument("--size", type=int, default=200)
ap.add_argument("--window", type=int, default=8)
ap.add_argument("--negative", type=int, default=24)
ap.add_argument("--iter", type=int, default=5)
ap.add_argument("--cpu-threads", default="1,8")
ap.add_argument("--gpu-threads", default="1,8,256,1024")
ap.add_argument("--variants", default="coherent,relaxed")
ap.add_argument("--repeats", type=int, default=120)
ap.add_argument("--tmp", default="/tmp/w2b_acc")
a = ap.parse_args()
os.makedirs(a.tmp, exist_ok=True)
corpus, questions = os.path.join(a.tmp, "planted.txt"), os.path.join(a.tmp, "questions.txt")
ntok = make_planted(corpus, questions, repeats=a.repeats)
REF = os.path.join(ROOT, "oracle", "_ref")
flags = ["-bitlevel", str(a.bitlevel), "-size", str(a.size), "-window", str(a.window), "-negative", str(a.negative),
         "-iter", str(a.iter), "-min-count", "5", "-binary", "1"]


def score(vec):
    with open(questions) as q:
        out = subprocess.run([os.path.join(REF, "compute_accuracy"), vec, "0", "0"], stdin=q, capture_output=True,
                             text=True).stdout
    return parse_accuracy(out)


def report(kind, threads, secs, vec, extra=None):
    r = {"kind": kind, "threads": threads, "secs": round(secs, 2), "tokens": ntok, "bitlevel": a.bitlevel,
         "size": a.size, "iter": a.iter}
    r.update(score(vec))
    if extra:
        r.update(extra)
    print(json.dumps(r), flush=True)


for th in [int(x) for x in a.cpu_threads.split(",") if x]:
    out = os.path.join(a.tmp, "ref_%d.bin" % th)
    t0 = time.time()
    subprocess.run([os.path.join(REF, "word2bits_stock"), "-train", corpus, "-output", out, "-threads", str(th)] + flags,
                   capture_output=True, text=True, check=True)
    report("reference-cpu", th, time.time() - t0, out)

for var in a.variants.split(","):
    relaxed = (var == "relaxed")
    for th in [int(x) for x in a.gpu_threads.split(",") if x]:
        out = os.path.join(a.tmp, "gpu%s_%d.bin" % (var, th))
        code = ("import sys,time; sys.path.insert(0,%r); import word2bits_amd as w; t0=time.time(); "
                "l=w.train_model(%r,%r,bitlevel=%d,size=%d,window=%d,negative=%d,threads=%d,iter=%d,min_count=5,binary=1,"
                "positions_per_launch=%d,relaxed_coherence=%r); print('LOSS', l[-1], time.time()-t0)" %
                (ROOT, corpus, out, a.bitlevel, a.size, a.window, a.negative, th, a.iter, 65536 if th < 64 else 4096,
                 relaxed))
        env = dict(os.environ)
        t0 = time.time()
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        if p.returncode != 0:
            print(json.dumps({"kind": "hip_" + var, "threads": th, "error": p.stderr[-400:]}), flush=True)
            continue
        report("hip_" + var, th, time.time() - t0, out, {"last_epoch_loss": float(p.stdout.split()[1])})
