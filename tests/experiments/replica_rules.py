"""Measurement, not a test: COMBINATION RULES of the replica exchange, swept in one process on ONE GPU through the phase API.

R replicas at the configs[1] shape live in this process (as in replicas8_cfg3.py); the "collective" is done here in torch on the
replicas' delta buffers, so that any rule -- per row or per element -- can be tried without rebuilding the library:
    d_r = W_r - base  (w2b_exchange_delta)   ->   combined = RULE(d_0 .. d_{R-1})   ->   W_r += combined - d_r, base += combined
                                                                                           (w2b_exchange_apply with scale 1)
Rules (--rules "a;b;c"):
  lib2                  the library's mode 2 as shipped in round 5 (hard threshold: mean of the contributors for rows with >= 32
                        expected updates per replica since the last exchange, sum otherwise) -- through the library's own path
  sum | mean            plain delta-sum / average over all replicas
  hard:T                the round-5 rule restated here: factor 1/R for rows with n >= T expected updates, 1 otherwise
  smooth:TU:TV          continuous saturation: a row that receives n updates per replica between two exchanges contracts towards
                        its equilibrium by rho = 1 - exp(-n / tau) in every replica; R replicas one after the other would have
                        contracted by 1 - (1 - rho)^R, so the SUM of the R deltas is scaled by
                            k(n) = (1 - exp(-R n / tau)) / (R (1 - exp(-n / tau)))          (-> 1 for n << tau, -> 1/R for n >> tau)
                        with tau = TU for rows of u (context rows) and TV for rows of v (target rows); n from the word counts.
  agree:P               per ELEMENT: mean + a^P (sum - mean) with the sign agreement a = |sum d_r| / sum |d_r|
  + suffix ",bf16"      the deltas are rounded to bfloat16 before they are combined (half the bytes on the links)
Schedules (--sync "every:K" | "geom:K0:KMAX"): exchange after every K launches, or at intervals that double from K0 to KMAX launches.
  python tests/experiments/replica_rules.py CORPUS.txt --positions 1024 --rules "lib2;smooth:32:32" [--out file.json]"""
import argparse, json, math, os, sys, time
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
ap.add_argument("--positions", type=int, default=1024)
ap.add_argument("--rules", default="lib2;smooth:32:32")
ap.add_argument("--sync", default="every:1")
ap.add_argument("--size", type=int, default=800)
ap.add_argument("--negative", type=int, default=24)
ap.add_argument("--window", type=int, default=8)
ap.add_argument("--bitlevel", type=int, default=1)
ap.add_argument("--sample", type=float, default=0.0, help="-sample of every trainer (the reference's default is 1e-3)")
ap.add_argument("--single", type=float, default=0.0, help="epoch loss of the single replica (skips that run)")
ap.add_argument("--single-validation", type=float, default=0.0)
ap.add_argument("--out", default="")
a = ap.parse_args()
dev = torch.device("cuda", 0)
t0 = time.time()
corpus = w2b.Corpus(a.corpus, 5)
tokens, counts = corpus.tokens(), corpus.counts()
V, D, R = corpus.vocab_size, a.size, a.replicas
print("RR corpus: %d words, vocabulary %d  [%.1f s]" % (corpus.train_words, V, time.time() - t0), flush=True)
N_MODEL = 2 * V * D
XCHUNK = ((N_MODEL + 3) & ~3) if N_MODEL < (64 << 20) else (64 << 20)      # w2b_trainer.cpp xchg_setup
# expected updates of every row of [u || v] per centre word (w2b_trainer.cpp word_rates / xchg_saturated_prefix)
c64 = counts.astype(np.float64)
rate = np.concatenate([(a.window + 1) * c64 / c64.sum(), a.negative * c64 ** 0.75 / (c64 ** 0.75).sum() + c64 / c64.sum()])
rate_dev = torch.tensor(rate, dtype=torch.float32, device=dev)
is_v = torch.zeros(2 * V, dtype=torch.bool, device=dev); is_v[V:] = True
res = {"corpus_words": int(corpus.train_words), "replicas": R, "workers_total": a.workers, "positions": a.positions, "sync": a.sync, "runs": []}


def row_factor(rule, words):
    """per-row factor on the SUM of the deltas, [2 V] on the device (None: the rule is not a per-row factor)"""
    kind = rule[0]
    n = rate_dev * float(words)
    if kind == "sum":
        return torch.ones_like(n)
    if kind == "mean":
        return torch.full_like(n, 1.0 / R)
    if kind == "hard":
        return torch.where(n >= float(rule[1]), torch.full_like(n, 1.0 / R), torch.ones_like(n))
    if kind == "smooth":
        tau = torch.where(is_v, torch.full_like(n, float(rule[2])), torch.full_like(n, float(rule[1])))
        x = (n / tau).double().clamp_min(1e-9)
        return ((1 - torch.exp(-R * x)) / (R * (1 - torch.exp(-x)))).float()
    if kind in ("smoothx", "signsafe"):
        # smoothx:TU:TV:FLOOR:HOTK -- the exponential saturation curve, never below FLOOR, and HOTK for the rows the single replica
        # would keep per-XCD copies of (rate >= 2^-7 per centre word at 1024 workers x 800 floats); signsafe uses it as its safe value
        tau = torch.where(is_v, torch.full_like(n, float(rule[2])), torch.full_like(n, float(rule[1])))
        x = (n / tau).double().clamp_min(1e-9)
        k = ((1 - torch.exp(-R * x)) / (R * (1 - torch.exp(-x)))).float()
        k = torch.maximum(k, torch.full_like(k, float(rule[3])))
        hotk = float(rule[4])
        if hotk > 0:
            k = torch.where(rate_dev >= 2.0 ** -7, torch.full_like(k, hotk), k)
        return k
    if kind == "table":
        # measured curves (tests/experiments/replica_truth.py): per table k(log2 n) piecewise linear, times a multiplier by the
        # row's rate log2(n / words), and a constant for the rows the single replica would give per-XCD copies (rate >= hot)
        spec = json.load(open(rule[1]))
        nn = n.double().clamp_min(2.0 ** -20).log2().cpu().numpy()
        rr = rate_dev.double().clamp_min(2.0 ** -40).log2().cpu().numpy()
        k = np.ones(2 * V)
        for tab, sl in (("u", slice(0, V)), ("v", slice(V, 2 * V))):
            t = spec[tab]
            kk = np.interp(nn[sl], [p[0] for p in t["k"]], [p[1] for p in t["k"]])
            kk = kk * np.interp(rr[sl], [p[0] for p in t["rate"]], [p[1] for p in t["rate"]])
            kk = np.where(rr[sl] >= t["hot_log2r"], t["hot_k"], kk)
            k[sl] = kk
        scale = float(rule[2]) if len(rule) > 2 else 1.0
        return torch.tensor(np.clip(k * scale, 0.0, 1.0), dtype=torch.float32, device=dev)
    return None


def chunk_factor(k_rows, off, length):
    r0, r1 = off // D, (off + length - 1) // D
    return k_rows[r0:r1 + 1].repeat_interleave(D)[off - r0 * D: off - r0 * D + length]


stats = {}


def exchange(ts, rule, words, bf16, first):
    if rule[0] == "lib2":                                  # the shipped path (counts + the library's own threshold)
        begun = [t.exchange_begin() for t in ts]
        cnts = [t.device_tensor(*t.exchange_counts()) for t in ts]
        total = torch.stack(cnts).sum(0)
        if first:
            stats["rows_touched_by_one_replica"] = float((cnts[0] > 0).float().mean())
            stats["rows_touched_by_any_replica"] = float((total > 0).float().mean())
        for b in cnts:
            b.copy_(total)
        torch.cuda.synchronize()
        for c in range(begun[0][0]):
            bufs = [t.device_tensor(*t.exchange_delta(c)) for t in ts]
            total = torch.stack(bufs).sum(0)
            for b in bufs:
                b.copy_(total)
            torch.cuda.synchronize()
            for t in ts:
                t.exchange_apply(c, 1.0)
    else:
        begun = [t.exchange_begin() for t in ts]
        k_rows = row_factor(rule, words)
        for c in range(begun[0][0]):
            bufs = [t.device_tensor(*t.exchange_delta(c)) for t in ts]
            st = torch.stack(bufs)
            if bf16:
                st = st.bfloat16().float()
            s = st.sum(0)
            if first and c == 0:
                stats["elements_touched_by_one_replica_chunk0"] = float((bufs[0] != 0).float().mean())
            if rule[0] == "signsafe":
                # per ELEMENT: the safe step (smoothx factor) decides the quantized value; where the big step (BIG x sum) lands in the
                # same quantization cell -- the same sign at one bit -- it is taken instead: the master keeps the inertia a single
                # shared model would have accumulated, the forward values are the safe rule's
                big = float(rule[5])
                base_c = ts[0].model_tensor()[c * XCHUNK: c * XCHUNK + s.numel()] - bufs[0]
                safe = s * chunk_factor(k_rows, c * XCHUNK, s.numel())
                bigs = s * big
                same = torch.signbit(base_c + safe) == torch.signbit(base_c + bigs)
                if first and c == 0:
                    stats["elements_taking_the_big_step_chunk0"] = float(same.float().mean())
                s = torch.where(same, bigs, safe)
                del base_c, safe, bigs, same
            elif k_rows is not None:
                s.mul_(chunk_factor(k_rows, c * XCHUNK, s.numel()))
            elif rule[0] == "agree":
                absum = st.abs().sum(0).clamp_min(1e-30)
                agree = (s.abs() / absum).pow_(float(rule[1]))
                mean = s / R
                s = mean + agree * (s - mean)
            else:
                raise SystemExit("unknown rule %r" % (rule,))
            del st
            for b in bufs:
                b.copy_(s)
            torch.cuda.synchronize()
            for t in ts:
                t.exchange_apply(c, 1.0)
    gw = sum(b[1] for b in begun)
    for t in ts:
        t.exchange_end(gw)


def sync_points(spec):
    """yields True at the launches after which an exchange happens"""
    kind, *args = spec.split(":")
    if kind == "every":
        k = int(args[0]); n = 0
        while True:
            n += 1
            yield n % k == 0, k
    else:                                                  # geom:K0:KMAX -- intervals K0, K0, 2 K0, 4 K0, ... capped at KMAX launches
        k, kmax = int(args[0]), int(args[1]); since = 0; done_at_k = 0
        while True:
            since += 1
            if since >= k:
                yield True, k
                since = 0; done_at_k += 1
                if done_at_k >= 2 and k < kmax:
                    k = min(2 * k, kmax); done_at_k = 0
            else:
                yield False, k


_val = {}


def validation_loss(t):
    """log-likelihood of a FIXED sample of (centre, 9 context words, 24 negatives) tuples under the trainer's final model:
    w2b_train_tuples with alpha = 0 computes the loss terms of ref :480-483 and changes nothing (g = 0).  What the replicas'
    final model is worth beside the single replica's, without the staleness that the on-line epoch loss also counts."""
    if not _val:
        rng = np.random.default_rng(99)
        nv = 200_000
        cdf = np.cumsum(c64[1:] / c64[1:].sum())
        draw = lambda m: (np.searchsorted(cdf, rng.random(m)) + 1).clip(1, V - 1).astype(np.int32)
        pw = c64[1:] ** 0.75
        cdfn = np.cumsum(pw / pw.sum())
        _val["center"] = draw(nv)
        _val["ctx"] = draw(nv * 9)
        _val["off"] = (np.arange(nv + 1) * 9).astype(np.int32)
        neg = (np.searchsorted(cdfn, rng.random(nv * a.negative)) + 1).clip(1, V - 1).astype(np.int32).reshape(nv, a.negative)
        neg[neg == _val["center"][:, None]] = -1
        _val["neg"] = neg
    t.synchronize()
    return t.train_tuples(_val["center"], _val["off"], _val["ctx"], _val["neg"], 0.0)


def run(Rn, rule, bf16=False):
    per = a.workers // Rn
    starts, ov = corpus.shards(a.workers)
    quota = corpus.train_words // a.workers
    ts = []
    for r in range(Rn):
        t = w2b.Trainer(V, D, a.window, a.negative, a.bitlevel, num_threads=per, iter=1, sample=a.sample, train_words=corpus.train_words,
                        compute_loss=True, worker_offset=r * per, total_threads=a.workers)
        t.init_net()
        t.set_vocab_counts(counts, 100_000_000)
        st = starts[r * per:(r + 1) * per]
        if Rn > 1:
            lo, hi, more = replicas.replica_token_slice(tokens, st, quota)
            t.set_corpus_slice(tokens[lo:hi], more)
            t.set_shards(st - lo, ov[r * per:(r + 1) * per])
            t.exchange_init()
        else:
            t.set_corpus(tokens)
            t.set_shards(st, ov)
        t.epoch_begin()
        ts.append(t)
    sched = sync_points(a.sync)
    launches = exchanges = 0
    since = 0
    while True:
        for t in ts:
            t.train_step(a.positions)
        launches += 1; since += 1
        done = all(t.epoch_poll(0)[0] for t in ts)
        due, _ = next(sched)
        if Rn > 1 and (done or due):
            exchange(ts, rule, since * a.positions * per, bf16, exchanges == 0)
            exchanges += 1; since = 0
        if done:
            break
    loss = sum(t.epoch_status()[3] for t in ts)
    val = validation_loss(ts[0])
    for t in ts:
        t.close()
    return loss, launches, exchanges, val


if a.single:
    one = a.single
    if a.single_validation:
        res["single_replica_validation"] = a.single_validation
else:
    t0 = time.time()
    one, l1, _, val1 = run(1, ("sum",))
    print("RR 1 replica x %d workers: loss %.0f, validation %.0f, %d launches  [%.0f s]" % (a.workers, one, val1, l1, time.time() - t0), flush=True)
    res["single_replica_validation"] = val1
res["single_replica_loss"] = one
for spec in a.rules.split(";"):
    spec = spec.strip()
    if not spec:
        continue
    bf16 = spec.endswith(",bf16")
    rule = tuple((spec[:-5] if bf16 else spec).split(":"))
    t0 = time.time()
    stats.clear()
    try:
        loss, launches, exchanges, val = run(R, rule, bf16)
    except Exception as e:                                  # keep sweeping
        print("RR rule %s failed: %r" % (spec, e), flush=True)
        continue
    dev_pct = 100 * (loss - one) / abs(one)
    rec = {"rule": spec, "loss": loss, "deviation_pct": dev_pct, "validation": val, "launches": launches, "exchanges": exchanges,
           "words_per_replica_per_launch": a.positions * (a.workers // R), **stats}
    res["runs"].append(rec)
    vdev = 100 * (val - res["single_replica_validation"]) / abs(res["single_replica_validation"]) if res.get("single_replica_validation") else float("nan")
    print("RR %d replicas, sync %-12s positions %5d, rule %-26s: loss %.0f (%+.2f %% vs 1 replica), final model on a fixed sample %.0f (%+.2f %%), %d launches, %d exchanges %s [%.0f s]" % (
        R, a.sync, a.positions, spec, loss, dev_pct, val, vdev, launches, exchanges, json.dumps(stats), time.time() - t0), flush=True)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)
corpus.close()
