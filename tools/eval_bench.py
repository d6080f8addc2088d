#!/usr/bin/env python3
"""Measures the evaluator's score kernel (w2b_kernels_eval.hip) on the GPU box and the reference evaluator beside it.

  python tools/eval_bench.py [--vocab 60238 --dim 200 --questions 19544 --kind 1bit --cpu-questions 24]

Prints one JSON line: questions/s end to end (top1 call: upload + query build + scan + download), the score
kernel's multiply-add rate from HIP events, its fraction of the vector-ALU peak for that arithmetic mode
(packed fp32: 39.3 T multiply-adds/s unfused, 78.6 T/s fused), and the unmodified reference evaluator
(oracle/_ref/compute_accuracy, one thread -- the program is single-threaded) on a bounded sample of questions.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import word2bits_amd as w2b                      # noqa: E402
from w2b_testlib import write_vectors_file, ref_binary   # noqa: E402

PEAK_MAC = {True: 78.6e12, False: 39.3e12}       # 256 CU x 4 SIMD x 16 lanes x 2.4 GHz x (2 | 1) per packed op pair


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vocab", type=int, default=60238)     # text8' vocabulary (reference README.md:122-131)
    ap.add_argument("--dim", type=int, default=200)
    ap.add_argument("--questions", type=int, default=19544)  # questions-words.txt
    ap.add_argument("--kind", default="1bit", choices=["1bit", "fp"])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-questions", type=int, default=24)
    a = ap.parse_args()
    rng = np.random.default_rng(3)
    V, D, Q = a.vocab, a.dim, a.questions
    M = ((rng.integers(0, 2, (V, D)) * 2 - 1).astype(np.float32) / np.float32(3)) if a.kind == "1bit" \
        else rng.standard_normal((V, D)).astype(np.float32)
    names = [("w%d" % i).encode() for i in range(V)]
    tmp = tempfile.mkdtemp()
    path = write_vectors_file(os.path.join(tmp, "v.bin"), names, M)
    b = rng.integers(0, V, (3, Q)).astype(np.int32)
    out = {"workload": "analogy scan: %d questions x %d rows x %d dims (%s)" % (Q, V, D, a.kind)}
    answers = {}
    for fused in (True, False):
        t0 = time.time()
        ev = w2b.Evaluator(path, 0, 0, fused=fused)
        load_s = time.time() - t0
        ev.top1(*b)
        ev.timing()
        t0 = time.time()
        for _ in range(a.reps):
            best, _ = ev.top1(*b)
        wall = (time.time() - t0) / a.reps
        ms, launches, macs = ev.timing()
        answers[fused] = best
        rate = macs / (ms * 1e-3)
        out["fused" if fused else "unfused"] = {
            "questions_per_s": Q / wall, "top1_call_ms": wall * 1e3, "load_s": load_s,
            "kernel_ms": ms / launches, "mac_per_s": rate, "peak_mac_per_s": PEAK_MAC[fused],
            "frac": rate / PEAK_MAC[fused]}
        ev.close()
    out["answers_differ_between_modes"] = int((answers[True] != answers[False]).sum())
    exe = ref_binary("compute_accuracy")
    if exe and a.cpu_questions > 0:
        n = a.cpu_questions
        qs = ": s\n" + "".join("w%d w%d w%d w%d\n" % (b[0, i], b[1, i], b[2, i], b[2, i]) for i in range(n))
        t0 = time.time()
        subprocess.run([exe, path, "0", "0"], input=b"", capture_output=True)
        t_load = time.time() - t0
        t0 = time.time()
        subprocess.run([exe, path, "0", "0"], input=qs.encode(), capture_output=True)
        t_all = time.time() - t0
        out["cpu_baseline"] = {"kind": "reference", "cores": 1, "questions_per_s": n / max(t_all - t_load, 1e-9),
                               "sample": "%d questions, load time (%.1f s) subtracted" % (n, t_load)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
