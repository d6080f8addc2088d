"""-m gpu: the ./word2bits command line end to end (ingest -> GPU epochs -> export -> save) against
the committed golden output files of the unmodified reference (tests/golden/)."""
import json
import os
import subprocess

import numpy as np
import pytest

from w2b_testlib import GOLDEN, ROOT, read_vectors

pytestmark = pytest.mark.gpu
META = json.load(open(os.path.join(GOLDEN, "golden.json")))
CORPUS = os.path.join(GOLDEN, "corpus_small.txt")
CLI = os.path.join(ROOT, "word2bits")


def run_cli(out, flags, threads=1, extra=()):
    args = [CLI, "-train", CORPUS, "-output", out, "-threads", str(threads)]
    for k, v in flags.items():
        args += ["-" + k.replace("_", "-"), str(v)]
    r = subprocess.run(args + list(extra), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-500:]
    return r.stdout


@pytest.mark.parametrize("name", ["b1_iter0", "b0_iter0_text"])
def test_cli_iter0_is_byte_identical_to_reference(gpu, name, tmp_path):
    """-iter 0: vocabulary order, InitNet (LCG), quantize(u+v) export and the writer -- no training,
    so the whole file must equal the reference's (SURVEY 8c rung 2)."""
    out = str(tmp_path / "o.vec")
    run_cli(out, META[name]["flags"])
    assert open(out, "rb").read() == open(os.path.join(GOLDEN, name + ".vec"), "rb").read()


@pytest.mark.parametrize("name", ["b1_d8", "b0_d12", "b2_d10_text", "b4_d8_reg", "b8_d8_nosample"])
def test_cli_single_thread_training_tracks_reference(gpu, name, tmp_path):
    flags = META[name]["flags"]
    out = str(tmp_path / "o.vec")
    txt = run_cli(out, flags)
    assert "Vocab size: %d" % META[name]["vocab_size"] in txt
    assert "Words in train file: %d" % META[name]["train_words"] in txt
    words, M = read_vectors(out, flags["binary"])
    gw, G = read_vectors(os.path.join(GOLDEN, name + ".vec"), flags["binary"])
    assert words == gw and M.shape == G.shape
    bl = flags["bitlevel"]
    if bl == 1:
        assert set(np.unique(M.view(np.uint32)).tolist()) <= {0x3EAAAAAB, 0xBEAAAAAB}   # legal levels only
    if bl == 2:
        assert set(np.unique(np.abs(M)).tolist()) <= {0.25, 0.75}
    if bl >= 4:
        steps = 1 << (bl - 1)
        assert np.array_equal(M * steps, np.round(M * steps)) and np.abs(M).max() <= 1.0
    # ~8000 positions on ~50 rows at D<=12: chaotic, but most output values still coincide
    if bl == 0:
        assert np.abs(M - G).mean() < 5e-3
    else:
        assert np.mean(M == G) > 0.80
    import re
    got = [float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", txt)]
    for a, b in zip(got, META[name]["epoch_loss"]):
        assert a == pytest.approx(b, rel=2e-2)


def test_cli_hogwild_many_workers(gpu, tmp_path):
    out = str(tmp_path / "o.vec")
    flags = dict(META["b1_d8"]["flags"])
    txt = run_cli(out, flags, threads=16, extra=["-save-every-epoch", "1", "-positions", "100"])
    assert os.path.exists(out + "_epoch0") and os.path.exists(out + "_epoch1")
    words, M = read_vectors(out, 1)
    assert len(words) == META["b1_d8"]["vocab_size"] and np.isfinite(M).all()
    assert txt.count("Starting epoch:") == 2 and txt.count("Epoch Loss:") == 2


def test_cli_threads_zero_fills_the_gpu(gpu, tmp_path):
    """-threads 0 (GPU extension): as many Hogwild workers as workgroups fit on the device -- capped so that every
    shard is at least five alpha periods long (a worker re-computes alpha only after >10000 of its own words,
    ref :379-393; with shorter shards the whole epoch would run at the starting alpha)."""
    import re
    out = str(tmp_path / "o.vec")
    flags = dict(META["b1_d8"]["flags"])
    txt = run_cli(out, flags, threads=0)                 # 4 000-token corpus: one worker
    m = re.search(r"Hogwild workers \(workgroups\): (\d+)", txt)
    assert m and int(m.group(1)) == max(1, META["b1_d8"]["train_words"] // 50000) == 1
    words, M = read_vectors(out, 1)
    assert np.isfinite(M).all() and len(words) == META["b1_d8"]["vocab_size"]
    # a corpus long enough for the whole device: the cap no longer binds, the resident workgroups decide
    rng = np.random.default_rng(0)
    big = str(tmp_path / "big.txt")
    ids = rng.integers(1, 2000, 12_000_000)
    with open(big, "w") as f:
        for i in range(0, len(ids), 1000):
            f.write(" ".join("w%d" % t for t in ids[i:i + 1000]) + "\n")
    r = subprocess.run([CLI, "-train", big, "-output", out, "-threads", "0", "-iter", "1", "-size", "64", "-window", "5",
                        "-negative", "5", "-binary", "1", "-min-count", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-300:]
    n = int(re.search(r"Hogwild workers \(workgroups\): (\d+)", r.stdout).group(1))
    tw = int(re.search(r"Words in train file: (\d+)", r.stdout).group(1))
    assert 200 <= n <= tw // 50000                        # (the cap: at least 50 000 words per worker and epoch)
    alphas = [float(a) for a in re.findall(r"Alpha: ([0-9.]+)", r.stdout)]
    assert alphas and min(alphas) < 0.04                  # the learning rate really decays inside the epoch
