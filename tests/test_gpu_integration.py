"""-m gpu: the boundary claim of INTEGRATION.md, demonstrated.  oracle/_ref/word2bits_hipseam is the REFERENCE program
(its own main(), flag parsing, vocabulary code, save loops, stdout) with only the thread fan-out of TrainModel replaced by
calls into libword2bits_hip.so (oracle/make_integration_build.py applies the INTEGRATION.md patch to a scratch copy of
the reference source; built where /root/reference is mounted, shipped to the GPU box as a binary).  With the parity mode
on (W2B_SEAM_EXACT=1 = w2b_config.exact_reduction) and -threads 1 it must write, byte for byte, the files the UNMODIFIED
reference wrote (tests/golden/*.vec) -- the C ABI is a sufficient seam."""
import json
import os
import re
import subprocess

import pytest

from w2b_testlib import GOLDEN, ref_binary

pytestmark = pytest.mark.gpu
META = json.load(open(os.path.join(GOLDEN, "golden.json")))
CORPUS = os.path.join(GOLDEN, "corpus_small.txt")


@pytest.mark.parametrize("name", sorted(META))
def test_patched_reference_writes_the_reference_files(gpu, name, tmp_path):
    exe = ref_binary("word2bits_hipseam")
    if not exe:
        pytest.skip("oracle/_ref/word2bits_hipseam not built (needs /root/reference at build time)")
    flags = META[name]["flags"]
    out = str(tmp_path / "o.vec")
    args = [exe, "-train", CORPUS, "-output", out, "-threads", "1"]
    for k, v in flags.items():
        args += ["-" + k.replace("_", "-"), str(v)]
    r = subprocess.run(args, capture_output=True, text=True, env=dict(os.environ, W2B_SEAM_EXACT="1"))
    assert r.returncode == 0, r.stdout[-400:] + r.stderr[-400:]
    assert "Vocab size: %d" % META[name]["vocab_size"] in r.stdout
    assert open(out, "rb").read() == open(os.path.join(GOLDEN, name + ".vec"), "rb").read()
    got = [float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", r.stdout)]
    assert len(got) == len(META[name]["epoch_loss"])
    for a, b in zip(got, META[name]["epoch_loss"]):
        assert a == pytest.approx(b, rel=1e-4)          # device logf/expf differ from glibc by ulps


def test_patched_reference_hogwild_fast_path(gpu, tmp_path):
    """the same binary on the fast path (no parity mode), 8 Hogwild workers, per-epoch files: runs, legal 1-bit levels"""
    import numpy as np
    from w2b_testlib import read_vectors
    exe = ref_binary("word2bits_hipseam")
    if not exe:
        pytest.skip("oracle/_ref/word2bits_hipseam not built")
    out = str(tmp_path / "o.vec")
    r = subprocess.run([exe, "-train", CORPUS, "-output", out, "-threads", "8", "-size", "64", "-window", "5", "-negative", "5",
                        "-iter", "2", "-min-count", "3", "-binary", "1", "-save-every-epoch", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-400:] + r.stderr[-400:]
    assert r.stdout.count("Epoch Loss:") == 2 and os.path.exists(out + "_epoch0") and os.path.exists(out + "_epoch1")
    words, M = read_vectors(out, 1)
    assert len(words) == 60 and set(np.unique(M.view(np.uint32)).tolist()) <= {0x3EAAAAAB, 0xBEAAAAAB}
