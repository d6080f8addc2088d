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
