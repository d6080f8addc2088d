#!/usr/bin/env python3
"""Generates tests/golden/eval_* by RUNNING THE UNMODIFIED REFERENCE EVALUATOR (oracle/_ref/compute_accuracy =
the Makefile:6 build with FMA contraction, and oracle/_ref/compute_accuracy_nofma = -ffp-contract=off; both built
by oracle/Makefile from /root/reference/src/compute-accuracy.c).  Only works where the reference mount exists;
the fixtures are small and committed so the checks travel to machines without it.

Fixtures: two vector files (a 1-bit one whose dot products tie massively -- the strict-greater / first-row
tie-break and the summation order decide every answer -- and a full-precision one with an odd dimension, a zero
row (NaN after normalisation), duplicate words after upper-casing and over-long words), four question streams whose expected answers are drawn from the (near-)tied best rows
(seven sections, out-of-vocabulary words, EXIT token, no trailing newline, truncated last question, empty), and
the evaluator's stdout for several (bitlevel, threshold) arguments under both builds.
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from w2b_testlib import ref_binary   # noqa: E402


def write_vectors(path, names, M):
    with open(path, "wb") as f:
        f.write(b"%d %d\n" % M.shape)
        for n, row in zip(names, M):
            f.write(n + b" " + row.astype("<f4").tobytes() + b"\n")


def make_names(rng, V):
    names = [b"</s>"]
    while len(names) < V:
        n = "".join(rng.choice(list("abcdefghij"), size=rng.integers(2, 6)))
        if rng.random() < 0.15:
            n = n.capitalize()
        names.append(n.encode())
    names[7], names[90] = b"the", b"The"                      # same word after toupper: row 7 wins the search
    names[40] = b"x" * 50                                     # exactly max_w characters
    names[41] = b"y" * 57                                     # longer than max_w
    names[42] = b"caf\xc3\xa9"                                # bytes >= 0x80 pass through toupper unchanged
    return names


def near_ties(M, b1, b2, b3):
    """Rows whose float64 score is within 1e-6 of the best one: the candidates between which the evaluator's
    float32 summation order and its strict-greater / first-row rule decide (computed independently of the oracle)."""
    with np.errstate(all="ignore"):
        Mn = M.astype(np.float64) / np.sqrt((M.astype(np.float64) ** 2).sum(1, keepdims=True))
    d = Mn @ (Mn[b2] - Mn[b1] + Mn[b3])
    d[[b1, b2, b3]] = -np.inf
    d[np.isnan(d)] = -np.inf
    return np.flatnonzero(d >= d.max() - 1e-6)


def make_questions(rng, names, M, n_sections, per_section, oov_rate, trailing_newline=True):
    lines = []
    for s in range(n_sections):
        lines.append(": section-%d" % s)
        for _ in range(per_section):
            ids = rng.integers(1, len(names), 4)
            if rng.random() < 0.8:                             # 4th word: one of the (near-)tied best rows
                ids[3] = rng.choice(near_ties(M[:len(names)], *ids[:3]))
            ws = [names[i].decode("latin1") for i in ids]
            if rng.random() < oov_rate:
                ws[rng.integers(0, 4)] = "notaword"
            if rng.random() < 0.3:
                ws = [w.lower() if rng.random() < .5 else w.upper() for w in ws]
            lines.append(" ".join(ws))
    txt = "\n".join(lines)
    return (txt + ("\n" if trailing_newline else "")).encode("latin1")


def main():
    rng = np.random.default_rng(20240924)
    V1, D1, V2, D2 = 260, 24, 400, 37
    n1, n2 = make_names(rng, V1), make_names(rng, V2)
    M1 = (rng.integers(0, 2, (V1, D1)) * 2 - 1).astype(np.float32) / np.float32(3)
    M2 = rng.standard_normal((V2, D2)).astype(np.float32)
    M2[11] = 0                                                 # len 0 -> the row becomes NaN (ref :109-110)
    M2[12] *= np.float32(1e-3)
    write_vectors(os.path.join(HERE, "eval_1bit.bin"), n1, M1)
    write_vectors(os.path.join(HERE, "eval_fp.bin"), n2, M2)
    q = {
        "eval_q_1bit.txt": make_questions(rng, n1[:200], M1, 7, 40, 0.15),
        "eval_q_fp.txt": make_questions(rng, n2[:300], M2, 8, 25, 0.15),
        "eval_q_noeol.txt": make_questions(rng, n1[:200], M1, 2, 12, 0.1, trailing_newline=False)
                            + b"\nEXIT tail\nab cd",         # EXIT acts as a section break; truncated question
        "eval_q_empty.txt": b"",
    }
    for name, data in q.items():
        open(os.path.join(HERE, name), "wb").write(data)
    cases = [("eval_1bit.bin", 0, 0), ("eval_1bit.bin", 1, 0), ("eval_1bit.bin", 0, 100),
             ("eval_fp.bin", 0, 0), ("eval_fp.bin", 1, 0), ("eval_fp.bin", 2, 0), ("eval_fp.bin", 4, 0),
             ("eval_fp.bin", 3, 150)]
    golden = []
    for build in ("compute_accuracy", "compute_accuracy_nofma"):
        exe = ref_binary(build)
        assert exe, "build oracle/_ref first (make -C oracle ref)"
        for vec, bitlevel, thr in cases:
            for qn in q:
                if qn in ("eval_q_1bit.txt", "eval_q_fp.txt") and qn[7:-4] not in vec:
                    continue
                r = subprocess.run([exe, os.path.join(HERE, vec), str(bitlevel), str(thr)],
                                   stdin=open(os.path.join(HERE, qn), "rb"), capture_output=True)
                golden.append({"build": build, "vectors": vec, "bitlevel": bitlevel, "threshold": thr,
                               "questions": qn, "stdout": r.stdout.decode("latin1")})
        r0 = subprocess.run([exe], capture_output=True)
        r1 = subprocess.run([exe, os.path.join(HERE, "no_such_file.bin")], capture_output=True, stdin=subprocess.DEVNULL)
        golden.append({"build": build, "cli": "usage", "stdout": r0.stdout.decode(), "returncode": r0.returncode})
        golden.append({"build": build, "cli": "notfound", "stdout": r1.stdout.decode(), "returncode": r1.returncode})
    with open(os.path.join(HERE, "eval_golden.json"), "w") as f:
        json.dump(golden, f, indent=0)
    runs = [g for g in golden if "vectors" in g]
    n_diff = sum(a["stdout"] != b["stdout"] for a, b in zip(runs[:len(runs) // 2], runs[len(runs) // 2:]))
    print("wrote %d transcripts; FMA vs no-FMA builds differ on %d" % (len(golden), n_diff))


if __name__ == "__main__":
    main()
