"""`bench.py --gpus N` produces N ranks or no line (round-3 review: --gpus was parsed and never read, so a bare
`python bench.py --gpus 8` would have reported n_gpus: 1).  CPU tests through the W2B_BENCH_DRY hook: the rendezvous
and the timing protocol run for real (torch.distributed.run, gloo), the GPU work is left out."""
import json
import os
import subprocess
import sys

from w2b_testlib import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, cwd=ROOT, timeout=timeout)
    return r, [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bare_gpus_2_starts_two_ranks():
    r, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"W2B_BENCH_DRY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout[-500:]            # one line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["value"] is None and "dry_run" in d
    assert "starting 2 ranks" in r.stderr


def test_bare_gpus_1_is_one_rank():
    r, lines = _run(["--gpus", "1", "--steps", "2", "--warmup", "0"], {"W2B_BENCH_DRY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1


def test_rendezvous_of_another_size_is_refused():
    # a rank of a 4-rank job asked for --gpus 2 (and the other way round): exit 2, nothing on stdout
    for gpus, world in (("2", "4"), ("8", "1"), ("1", "2")):
        r, lines = _run(["--gpus", gpus, "--steps", "1", "--warmup", "0"],
                        {"W2B_BENCH_DRY": "1", "WORLD_SIZE": world, "RANK": "0", "LOCAL_RANK": "0"}, timeout=60)
        assert r.returncode == 2, (gpus, world, r.returncode, r.stderr[-500:])
        assert not lines and "refusing" in r.stderr


def test_cpu_legs_of_bench_read_the_reference_programs_stdout():
    """bench.py's CPU legs watch the unmodified reference program through its unbuffered stdout ('Starting epoch' -> 'Epoch Loss'
    for a whole epoch, the progress line's percentage for a bounded sample): both modes on a corpus of a few seconds.  Needs
    oracle/_ref (built where /root/reference is mounted); the legs themselves only ever run in bench.py's cpu_baseline part."""
    import argparse
    import importlib.util
    import pytest
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "word2bits_stock")):
        pytest.skip("oracle/_ref/word2bits_stock not built here")
    spec = importlib.util.spec_from_file_location("bench", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = argparse.Namespace(vocab=5000, cpu_tokens=400_000, bitlevel=1, dim=100, window=8, negative=24)
    path = bench.write_cpu_corpus(a)
    try:
        whole = bench.cpu_baseline_reference(a, path)
        assert whole["kind"] == "reference" and whole["cores"] == (os.cpu_count() or 1) and whole["value"] > 1000
        assert "424995 tokens" in whole["sample"] or "tokens =" in whole["sample"]
        r = bench.RefProbe(path, bench.ref_flags(a), 1, sample_seconds=4.0).result()
        assert r and 1000 < r["words_per_s"] < 5e6
        assert r["whole_epoch"] or r["train_s"] >= 2.0          # (a fast host may finish the epoch inside the sample)
    finally:
        os.remove(path)
        os.rmdir(os.path.dirname(path))
