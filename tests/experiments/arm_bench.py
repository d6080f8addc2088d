"""Measurement, not a test: throughput of the form-(i) training launch at the headline shape (or --dim/--vocab/...) under
several ARMS of w2b_tuning knobs, all in ONE process on ONE box: the synthetic stream is drawn once, every arm gets a
fresh trainer, `--rounds` interleaved rounds (A B C A B C ...) so that clock / thermal drift does not favour an arm.

  python tests/experiments/arm_bench.py --arms "default:;loss:loss=1;late:hot_late=1;fresh:fresh_rank_u=2000"
An arm is name:key=value,key=value ; keys are w2b_tuning fields plus loss / window_cache / row_groups / workers / bitlevel."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import word2bits_amd as w2b
from word2bits_amd import replicas

ap = argparse.ArgumentParser()
ap.add_argument("--arms", default="default:")
ap.add_argument("--vocab", type=int, default=400_000)
ap.add_argument("--dim", type=int, default=800)
ap.add_argument("--window", type=int, default=8)
ap.add_argument("--negative", type=int, default=24)
ap.add_argument("--bitlevel", type=int, default=1)
ap.add_argument("--tokens", type=int, default=30_000_000)
ap.add_argument("--batch", type=int, default=1 << 20)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--out", default="")
ap.add_argument("--ids", choices=["zipf", "uniform"], default="zipf")
ap.add_argument("--zipf-shift", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda", 0)
V, D, W, K = a.vocab, a.dim, a.window, a.negative
gen = torch.Generator(device=dev); gen.manual_seed(1000)
w = torch.ones(V - 1, dtype=torch.float64, device=dev) if a.ids == "uniform" else 1.0 / (torch.arange(1, V, dtype=torch.float64, device=dev) + a.zipf_shift)
cdf = torch.cumsum(w, 0)
cdf = cdf / cdf[-1]
stream = torch.empty(a.tokens, dtype=torch.int32, device=dev)
for o in range(0, a.tokens, 1 << 24):
    m = min(1 << 24, a.tokens - o)
    stream[o:o + m] = (torch.searchsorted(cdf, torch.rand(m, dtype=torch.float64, device=dev, generator=gen)) + 1).clamp_(max=V - 1).to(torch.int32)
stream[999::1000] = 0
counts = torch.bincount(stream.long(), minlength=V).clamp_(min=1).cpu().numpy().astype(np.int64)
train_words = int(counts.sum())
bpw = 8 * D * (W + 1 + K + 1) + 4 * (1 + W + 1 + K)

arms = []
for part in a.arms.split(";"):
    if part.strip():
        name, _, kv = part.partition(":")
        d = {}
        for x in kv.split(","):
            if x.strip():
                k, _, v = x.partition("=")
                d[k.strip()] = int(v)
        arms.append((name.strip(), d))


def one(name, kw):
    kw = dict(kw)
    loss = bool(kw.pop("loss", 0))
    wc = kw.pop("window_cache", -1)
    wcache = None if wc < 0 else bool(wc)
    rg = kw.pop("row_groups", -1)
    rgroups = None if rg < 0 else bool(rg)
    workers = kw.pop("workers", 0)
    bitlevel = kw.pop("bitlevel", a.bitlevel)
    if workers <= 0:
        probe = w2b.Trainer(V, D, W, K, bitlevel, num_threads=1, device=0, sample=0.0, train_words=train_words, window_cache=wcache, row_groups=rgroups, compute_loss=loss)
        probe.set_vocab_counts(counts, 0)
        workers = probe.suggested_threads()
        probe.close()
    t = w2b.Trainer(V, D, W, K, bitlevel, num_threads=workers, iter=1, alpha=0.05, sample=0.0, reg=0.0, train_words=train_words,
                    compute_loss=loss, device=0, window_cache=wcache, row_groups=rgroups, **kw)
    t.init_net()
    t.set_vocab_counts(counts, 100_000_000)
    t.set_corpus_device(stream.data_ptr(), a.tokens)
    t.set_shards(replicas.token_shard_starts(a.tokens, workers, 0, workers))
    t.epoch_begin()
    info = t.worker_kernel_info()
    kname = t.worker_kernel_name()
    positions = max(1, a.batch // workers)
    for _ in range(a.warmup):
        t.train_step(positions)
    t.synchronize(); t.timing_enable(True); t.timing_read()
    for _ in range(a.steps):
        t.train_step(positions)
    t.synchronize()
    ms, n = t.timing_read()
    t.close()
    wps = workers * positions / (ms / n / 1e3)
    return {"arm": name, "knobs": kw, "loss": loss, "workers": workers, "resident": bool(info[0]), "kernel": kname, "us_per_word_per_worker": 1e6 * workers / wps, "hot_rows": info[4], "Mwords_s": wps / 1e6,
            "frac": wps * bpw / 8e12, "launch_ms": ms / n}


res = {}
for r in range(a.rounds):
    for name, kw in arms:
        try:
            x = one(name, kw)
        except Exception as e:
            print("%-26s FAILED %r" % (name, e), flush=True)
            continue
        res.setdefault(name, []).append(x)
        print("round %d %-26s %7.2f Mw/s  frac %.3f  launch %.2f ms  workers %d %s hot %d  %.1f us/word/worker" % (
            r, name, x["Mwords_s"], x["frac"], x["launch_ms"], x["workers"], x["kernel"], x["hot_rows"], x["us_per_word_per_worker"]), flush=True)
print("== best of %d rounds" % a.rounds)
for name, xs in res.items():
    b = max(xs, key=lambda x: x["Mwords_s"])
    print("%-26s %7.2f Mw/s  frac %.3f  (all rounds: %s)" % (name, b["Mwords_s"], b["frac"], " ".join("%.3f" % x["frac"] for x in xs)))
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
