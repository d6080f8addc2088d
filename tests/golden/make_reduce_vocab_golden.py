#!/usr/bin/env python3
"""Generates tests/golden/reduce_vocab.json by RUNNING THE REFERENCE with one constant changed on its way into the
compiler (oracle/_ref/word2bits_nofma_hash3000: vocab_hash_size 30000000 -> 3000, oracle/Makefile), so that ReduceVocab
(ref :245-263) runs on corpora of a few thousand distinct words.  The corpora are regenerated from integers
(w2b_testlib.write_reduce_vocab_corpus); the fixture keeps, per corpus and flag set, the reference's 'Vocab size' /
'Words in train file' lines, its vocabulary in row order and the sha256 of its complete output file."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from w2b_testlib import read_vectors, ref_binary, write_reduce_vocab_corpus   # noqa: E402

FLAGS = {
    "iter0": dict(bitlevel=1, size=4, window=3, negative=4, iter=0, min_count=1, binary=1),
    "train": dict(bitlevel=2, size=8, window=3, negative=4, iter=1, min_count=3, binary=1),
}


def main():
    exe = ref_binary("word2bits_nofma_hash3000")
    assert exe, "build oracle/_ref first (make -C oracle ref)"
    meta = {}
    with tempfile.TemporaryDirectory() as tmp:
        for kind in ("zipf", "wipe"):
            corpus = write_reduce_vocab_corpus(os.path.join(tmp, kind + ".txt"), kind)
            meta[kind] = {"corpus_sha256": hashlib.sha256(open(corpus, "rb").read()).hexdigest()}
            for name, flags in FLAGS.items():
                out = os.path.join(tmp, "o.vec")
                args = [exe, "-train", corpus, "-output", out, "-threads", "1"]
                for k, v in flags.items():
                    args += ["-" + k.replace("_", "-"), str(v)]
                txt = subprocess.run(args, check=True, capture_output=True, text=True).stdout
                words, _ = read_vectors(out, True)
                meta[kind][name] = {
                    "flags": flags,
                    "vocab_size": int(re.search(r"Vocab size: (\d+)", txt).group(1)),
                    "train_words": int(re.search(r"Words in train file: (\d+)", txt).group(1)),
                    "row0": words[0],
                    "words_sha256": hashlib.sha256("\n".join(words).encode("latin1")).hexdigest(),
                    "output_sha256": hashlib.sha256(open(out, "rb").read()).hexdigest(),
                }
    with open(os.path.join(HERE, "reduce_vocab.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
