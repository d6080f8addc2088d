"""Measurement, not a test: BASELINE configs[3] on ONE GPU through the phase API -- 8 replicas at the configs[1] shape
(V = 400 K, size 800, negative 24, bitlevel 1), 128 workers each, a full exchange after every launch (--positions 0: the interval
./word2bits -gpus 8 picks, 1 / 32 of a replica's epoch between 32 K and 1 M centre words; 8192: 1 M words, what the xGMI links carry
beside a full-device launch), against the single replica with the same 1024 workers and the same launches
and against 8 replicas that meet at the end of the epoch only.  Also timed: the elementwise kernels of one full exchange
(k_xchg_delta, k_xchg_apply; the collective itself is a device-side sum here) beside one training launch, and printed: the bytes one
exchange puts on the links with the two link models (ring over one ~153 GB/s link; reduce-scatter + all-gather over all seven).
  python tests/experiments/replicas8_cfg3.py CORPUS.txt [--replicas 8] [--positions 0] [--out file.json]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.init()
import word2bits_amd as w2b
from word2bits_amd import replicas

ap = argparse.ArgumentParser()
ap.add_argument("corpus")
ap.add_argument("--replicas", type=int, default=8)
ap.add_argument("--workers", type=int, default=1024)
ap.add_argument("--positions", type=int, default=0)
ap.add_argument("--modes", default="2")
ap.add_argument("--out", default="")
a = ap.parse_args()
flags = dict(bitlevel=1, size=800, window=8, negative=24)
t0 = time.time()
corpus = w2b.Corpus(a.corpus, 5)
print("R8 corpus: %d words, vocabulary %d  [%.1f s]" % (corpus.train_words, corpus.vocab_size, time.time() - t0), flush=True)
tokens = corpus.tokens()
counts = corpus.counts()
if a.positions <= 0:                                      # the command line's rule (word2bits_main.cpp): 1 / 32 of a replica's epoch
    words = w2b.lib().w2b_suggested_exchange_words(int(corpus.train_words), a.replicas)
    a.positions = max(16, words // (a.workers // a.replicas))
res = {"corpus_words": int(corpus.train_words), "workers_total": a.workers, "positions": a.positions, "runs": []}


def timed(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3


def exchange(ts, mode, cost):
    begun = [t.exchange_begin() for t in ts]
    n_chunks = begun[0][0]
    scale = 1.0 / len(ts) if mode == 1 else 1.0          # mode 0 delta-sum, 1 average, 2 contributor mean of the saturated rows
    if mode == 2:
        cnts = [t.device_tensor(*t.exchange_counts()) for t in ts]
        total = torch.stack(cnts).sum(0)
        for b in cnts:
            b.copy_(total)
        torch.cuda.synchronize()
    for c in range(n_chunks):
        bufs = []
        for i, t in enumerate(ts):
            if i == 0:
                holder = {}
                cost["delta_ms"] += timed(lambda: holder.__setitem__("b", t.exchange_delta(c)))
                bufs.append(t.device_tensor(*holder["b"]))
            else:
                bufs.append(t.device_tensor(*t.exchange_delta(c)))
        total = torch.stack(bufs).sum(0)
        for b in bufs:
            b.copy_(total)
        torch.cuda.synchronize()
        for i, t in enumerate(ts):
            if i == 0:
                cost["apply_ms"] += timed(lambda: t.exchange_apply(c, scale))
            else:
                t.exchange_apply(c, scale)
    cost["exchanges"] += 1
    words = sum(b[1] for b in begun)
    for t in ts:
        t.exchange_end(words)


def run(R, sync_every, mode=2):
    per = a.workers // R
    starts, ov = corpus.shards(a.workers)
    quota = corpus.train_words // a.workers
    ts = []
    for r in range(R):
        t = w2b.Trainer(corpus.vocab_size, flags["size"], flags["window"], flags["negative"], flags["bitlevel"], num_threads=per, iter=1,
                        sample=0.0, train_words=corpus.train_words, compute_loss=True, worker_offset=r * per, total_threads=a.workers)
        t.init_net()
        t.set_vocab_counts(counts, 100_000_000)
        st = starts[r * per:(r + 1) * per]
        if R > 1:
            lo, hi, more = replicas.replica_token_slice(tokens, st, quota)
            t.set_corpus_slice(tokens[lo:hi], more)
            t.set_shards(st - lo, ov[r * per:(r + 1) * per])
            t.exchange_init()
        else:
            t.set_corpus(tokens)
            t.set_shards(st, ov)
        t.epoch_begin()
        ts.append(t)
    cost = {"delta_ms": 0.0, "apply_ms": 0.0, "exchanges": 0, "launch_ms": 0.0, "launches": 0}
    launches = 0
    while True:
        for i, t in enumerate(ts):
            if i == 0:
                cost["launch_ms"] += timed(lambda: t.train_step(a.positions)); cost["launches"] += 1
            else:
                t.train_step(a.positions)
        launches += 1
        done = all(t.epoch_poll(0)[0] for t in ts)
        if R > 1 and (done or (sync_every > 0 and launches % sync_every == 0)):
            exchange(ts, mode, cost)
        if done:
            break
    loss = sum(t.epoch_status()[3] for t in ts)
    kernel = ts[0].worker_kernel_name()
    for t in ts:
        t.close()
    return loss, launches, cost, kernel


t0 = time.time()
one, l1, c1, k1 = run(1, 0)
print("R8 1 replica x %d workers (%s kernel): loss %.0f, %d launches of %.1f ms  [%.0f s]" % (a.workers, k1, one, l1, c1["launch_ms"] / max(1, c1["launches"]), time.time() - t0), flush=True)
res["runs"].append({"replicas": 1, "scheme": "single replica", "loss": one, "launches": l1, "kernel": k1})
V, D = corpus.vocab_size, flags["size"]
model_gb = 2 * V * D * 4 / 1e9
for name, se in (("end of the epoch only", 0), ("full exchange after every launch", 1)):
    for mode in ([int(x) for x in a.modes.split(",")] if se else [2]):
        t0 = time.time()
        loss, launches, cost, kernel = run(a.replicas, se, mode)
        n = max(1, cost["exchanges"])
        words_between = a.positions * (a.workers // a.replicas)
        rec = {"replicas": a.replicas, "scheme": name, "mode": mode, "loss": loss, "deviation_pct": 100 * (loss - one) / abs(one), "launches": launches,
               "centre_words_per_replica_between_exchanges": words_between, "kernel": kernel,
               "launch_ms": cost["launch_ms"] / max(1, cost["launches"]),
               "xchg_delta_ms_per_exchange": cost["delta_ms"] / n, "xchg_apply_ms_per_exchange": cost["apply_ms"] / n,
               "xchg_elementwise_ms_per_exchange": (cost["delta_ms"] + cost["apply_ms"]) / n,
               "xchg_elementwise_share_of_a_launch": (cost["delta_ms"] + cost["apply_ms"]) / n / max(1e-9, cost["launch_ms"] / max(1, cost["launches"])),
               "model_GB": model_gb,
               "xchg_delta_GB": 4 * model_gb / 2 * 2 / 2 * 1.0,      # reads w, base; writes d, s  (4 x model bytes ... see below)
               }
        # bytes: delta reads w + base and writes d + s = 4 model-sized passes; apply reads w, base, d, s (+ counts) and writes w, base = 6
        rec["xchg_delta_GB"] = 4 * model_gb
        rec["xchg_apply_GB"] = 6 * model_gb
        # what one exchange puts on the links: a ring all-reduce moves 2 (R - 1) / R of the model over ONE ~153 GB/s xGMI link, a direct
        # reduce-scatter + all-gather 2 / R of it over each of the R - 1 links at once; beside the launch(es) of one interval -- of a
        # 128-worker replica here, and of a full device (1024 workers: 27 M words/s) in a real 8-GPU job
        R = a.replicas
        rec["link_bytes_per_exchange"] = model_gb * 1e9
        rec["link_ms_ring_one_link"] = 2 * (R - 1) / R * model_gb / 153 * 1e3
        rec["link_ms_direct_all_links"] = 2 * model_gb / R / 153 * 1e3
        rec["interval_ms_this_replica"] = rec["launch_ms"]
        rec["interval_ms_full_device"] = words_between / 27e6 * 1e3
        res["runs"].append(rec)
        print("R8 %d replicas x %d workers, %-34s mode %d: loss %.0f (%+.2f %% vs 1 replica), %d launches of %.1f ms (%d words/replica), "
              "exchange kernels %.1f + %.1f ms per exchange = %.0f %% of a launch; %.2f GB per exchange on the links: ring over one link %.1f ms, "
              "direct over all links %.1f ms, beside an interval of %.1f ms (this 128-worker replica) / %.1f ms (a full device)  [%.0f s]" % (
                  a.replicas, a.workers // a.replicas, name, mode, loss, rec["deviation_pct"], launches, rec["launch_ms"], words_between,
                  rec["xchg_delta_ms_per_exchange"], rec["xchg_apply_ms_per_exchange"], 100 * rec["xchg_elementwise_share_of_a_launch"],
                  model_gb, rec["link_ms_ring_one_link"], rec["link_ms_direct_all_links"], rec["interval_ms_this_replica"],
                  rec["interval_ms_full_device"], time.time() - t0), flush=True)
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
corpus.close()
