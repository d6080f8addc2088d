#!/usr/bin/env python3
"""Generates tests/golden/* by RUNNING THE UNMODIFIED REFERENCE (oracle/_ref/word2bits_nofma, built by
oracle/Makefile from /root/reference).  Only works where the reference mount exists; the outputs
are small and committed so the checks travel to machines without it.

Fixtures: a tiny corpus (committed as text), and for several flag sets the reference's complete
output file (binary or text) plus its 'Vocab size' / 'Words in train file' / 'Epoch Loss' lines.
"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from w2b_testlib import write_corpus, ref_binary   # noqa: E402

CASES = {
    # name: flags (all -threads 1: bit-deterministic)
    "b1_d8": dict(bitlevel=1, size=8, window=3, negative=4, iter=2, min_count=2, binary=1),
    "b0_d12": dict(bitlevel=0, size=12, window=5, negative=5, iter=2, min_count=3, binary=1),
    "b2_d10_text": dict(bitlevel=2, size=10, window=8, negative=24, iter=1, min_count=1, binary=0),
    "b4_d8_reg": dict(bitlevel=4, size=8, window=2, negative=3, iter=2, min_count=2, binary=1, reg=0.001),
    "b8_d8_nosample": dict(bitlevel=8, size=8, window=4, negative=3, iter=1, min_count=2, binary=1, sample=0),
    "b1_iter0": dict(bitlevel=1, size=16, window=5, negative=5, iter=0, min_count=1, binary=1),
    "b0_iter0_text": dict(bitlevel=0, size=6, window=5, negative=5, iter=0, min_count=4, binary=0),
}


def main():
    exe = ref_binary("word2bits_nofma")
    assert exe, "build oracle/_ref first (make -C oracle ref)"
    corpus = os.path.join(HERE, "corpus_small.txt")
    write_corpus(corpus, seed=7, vocab=60, n_tokens=4000, line_len=25)
    meta = {}
    for name, flags in CASES.items():
        out = os.path.join(HERE, name + ".vec")
        args = [exe, "-train", corpus, "-output", out, "-threads", "1"]
        for k, v in flags.items():
            args += ["-" + k.replace("_", "-"), str(v)]
        txt = subprocess.run(args, check=True, capture_output=True, text=True).stdout
        meta[name] = {
            "flags": flags,
            "vocab_size": int(re.search(r"Vocab size: (\d+)", txt).group(1)),
            "train_words": int(re.search(r"Words in train file: (\d+)", txt).group(1)),
            "epoch_loss": [float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", txt)],
        }
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(CASES), "fixtures")


if __name__ == "__main__":
    main()
