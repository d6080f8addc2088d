"""Planted-analogy corpus (test infrastructure): an offline stand-in for text8 + the Google analogy set
(no network; SURVEY.md finding 5 / Appendix C.6 idea, re-done with synthetic words so that no
reference data file is needed at run time).

S sections of P word pairs (a_i, b_i).  Every pair shares 4 pair-specific context tokens, every
section has 4 tokens for its "a side" and 4 for its "b side".  The corpus holds, per pair, R times
the two 7-token lines
    a_i + 3 of the 4 pair tokens + 3 of the 4 a-side tokens
    b_i + 3 of the 4 pair tokens + 3 of the 4 b-side tokens
shuffled.  Then b_j ~ a_j - a_i + b_i holds for pairs of one section, which is exactly what the
reference's compute_accuracy (src/compute-accuracy.c) scores from a questions file
": section" / "a_i b_i a_j b_j" lines.  Sections 1..5 count as "semantic", the rest as "syntactic"
(src/compute-accuracy.c:181-183).
"""
import numpy as np


def make_planted(corpus_path, questions_path, sections=14, pairs=24, repeats=120, seed=0, max_questions=None):
    rng = np.random.default_rng(seed)
    lines = []
    q = []
    for s in range(sections):
        side_a = ["sa%d_%d" % (s, k) for k in range(4)]
        side_b = ["sb%d_%d" % (s, k) for k in range(4)]
        q.append(": section%d" % s)
        for i in range(pairs):
            for j in range(pairs):
                if i != j:
                    q.append("a%d_%d b%d_%d a%d_%d b%d_%d" % (s, i, s, i, s, j, s, j))
        for i in range(pairs):
            ptok = ["p%d_%d_%d" % (s, i, k) for k in range(4)]
            for _ in range(repeats):
                for head, side in (("a%d_%d" % (s, i), side_a), ("b%d_%d" % (s, i), side_b)):
                    toks = [head] + list(rng.permutation(ptok)[:3]) + list(rng.permutation(side)[:3])
                    toks = [toks[0]] + list(rng.permutation(toks[1:]))
                    lines.append(" ".join(toks))
    order = rng.permutation(len(lines))
    with open(corpus_path, "w") as f:
        for k in order:
            f.write(lines[k])
            f.write("\n")
    if max_questions:
        q = q[:max_questions]
    with open(questions_path, "w") as f:
        f.write("\n".join(q) + "\n")
    return len(lines) * 7


def parse_accuracy(stdout):
    """last 'Total accuracy: x %   Semantic accuracy: y %   Syntactic accuracy: z %' + questions seen"""
    import re
    tot = re.findall(r"Total accuracy: ([\d.]+) %\s+Semantic accuracy: ([\d.naNA-]+) %\s+Syntactic accuracy: ([\d.naNA-]+) %", stdout)
    seen = re.search(r"Questions seen / total: (\d+) (\d+)", stdout)
    t = tot[-1] if tot else ("nan", "nan", "nan")
    f = lambda x: float(x) if x.replace(".", "").isdigit() else float("nan")
    return {"total": f(t[0]), "semantic": f(t[1]), "syntactic": f(t[2]),
            "seen": int(seen.group(1)) if seen else 0, "questions": int(seen.group(2)) if seen else 0}
