"""Measurement, not a test: what SHOULD the combination of R replicas' deltas be?  Measured, per row, against a truth run.

R replicas at the configs[1] shape live in this process (phase API, as replica_rules.py) and, beside them, ONE trainer with all
R x workers workers on the whole stream -- the single replica the 8-replica job is compared with.  At the start of every
exchange interval the truth trainer is set to the common model `base`; over the interval it trains exactly the words the R
replicas train (same global worker ids, same shards, same launches).  At the end
    d_r = W_r - base (replica r),   S = sum_r d_r,   T = W_truth - base
and for every row of [u || v] the least-squares factor on the sum,  k* = <T, S> / <S, S>,  is pooled over rows with a similar
number n of expected updates per replica (n = rate x words, log2 bins): k*(n) is the curve a per-row rule of the form
"factor(n) x sum" should follow.  Also pooled: how much of T a scaled S explains, <T,S>^2 / (<S,S><T,T>), and how aligned the
replicas' deltas are, <S,S> / sum_r <d_r,d_r> (1 = orthogonal, R = identical).
--apply oracle : every replica then ADOPTS the truth (combined = T): the epoch loss of the replicas under a perfect combination
                 rule -- what the interval alone costs;
--apply smooth:TU:TV | curve:FILE : the replicas go on with that rule (the truth trainer follows the merged model).
  python tests/experiments/replica_truth.py CORPUS.txt --positions 1024 --apply oracle --out curves.json"""
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
ap.add_argument("--every", type=int, default=1, help="launches per exchange interval")
ap.add_argument("--apply", default="oracle")
ap.add_argument("--size", type=int, default=800)
ap.add_argument("--negative", type=int, default=24)
ap.add_argument("--window", type=int, default=8)
ap.add_argument("--bitlevel", type=int, default=1)
ap.add_argument("--log-at", default="1,2,3,5,8,12,16,24,32,48,64,96,128", help="exchanges whose curves are printed")
ap.add_argument("--max-exchanges", type=int, default=0, help="stop after this many exchanges (0 = the whole epoch)")
ap.add_argument("--out", default="")
a = ap.parse_args()
dev = torch.device("cuda", 0)
corpus = w2b.Corpus(a.corpus, 5)
tokens, counts = corpus.tokens(), corpus.counts()
V, D, R = corpus.vocab_size, a.size, a.replicas
N_MODEL = 2 * V * D
XCHUNK = ((N_MODEL + 3) & ~3) if N_MODEL < (64 << 20) else (64 << 20)
c64 = counts.astype(np.float64)
kept_tot = c64[1:].sum()
rate = np.zeros(2 * V)
rate[1:V] = (a.window + 1) * c64[1:] / kept_tot
rate[V + 1:] = a.negative * c64[1:] ** 0.75 / (c64 ** 0.75).sum() + c64[1:] / kept_tot
rate_dev = torch.tensor(rate, dtype=torch.float32, device=dev)
is_v = torch.zeros(2 * V, dtype=torch.bool, device=dev); is_v[V:] = True
log_at = {int(x) for x in a.log_at.split(",") if x}
res = {"corpus_words": int(corpus.train_words), "replicas": R, "positions": a.positions, "every": a.every, "apply": a.apply, "curves": []}


def make(nw, offset, total, whole):
    t = w2b.Trainer(V, D, a.window, a.negative, a.bitlevel, num_threads=nw, iter=1, sample=0.0, train_words=corpus.train_words,
                    compute_loss=True, worker_offset=offset, total_threads=total)
    t.init_net()
    t.set_vocab_counts(counts, 100_000_000)
    starts, ov = corpus.shards(total)
    st = starts[offset:offset + nw]
    if whole:
        t.set_corpus(tokens)
        t.set_shards(st, ov[offset:offset + nw])
    else:
        lo, hi, more = replicas.replica_token_slice(tokens, st, corpus.train_words // total)
        t.set_corpus_slice(tokens[lo:hi], more)
        t.set_shards(st - lo, ov[offset:offset + nw])
        t.exchange_init()
    t.epoch_begin()
    return t


per = a.workers // R
ts = [make(per, r * per, a.workers, False) for r in range(R)]
truth = make(a.workers, 0, a.workers, True)
base = truth.model_tensor().clone()
S_full = torch.empty(N_MODEL, dtype=torch.float32, device=dev)
Q_full = torch.empty(N_MODEL, dtype=torch.float32, device=dev)          # sum_r d_r^2


def smooth_rows(words, tu, tv):
    n = rate_dev * float(words)
    tau = torch.where(is_v, torch.full_like(n, tv), torch.full_like(n, tu))
    x = (n / tau).double().clamp_min(1e-9)
    return ((1 - torch.exp(-R * x)) / (R * (1 - torch.exp(-x)))).float()


def curve_rows(words, spec):
    """spec: {"u": [[log2n, k], ...], "v": [...]} -- piecewise linear in log2 n"""
    n = (rate_dev * float(words)).clamp_min(1e-6).log2().cpu().numpy()
    k = np.ones(2 * V, np.float32)
    for tab, sl in (("u", slice(0, V)), ("v", slice(V, 2 * V))):
        pts = np.array(spec[tab], np.float64)
        k[sl] = np.interp(n[sl], pts[:, 0], pts[:, 1])
    return torch.tensor(k, device=dev)


def chunk_rows(k_rows, off, length):
    r0, r1 = off // D, (off + length - 1) // D
    return k_rows[r0:r1 + 1].repeat_interleave(D)[off - r0 * D: off - r0 * D + length]


def pooled(words, T_full):
    """curves per table: log2 bins of n -> rows, pooled k*, explained fraction, alignment"""
    n = rate_dev * float(words)
    Sr = S_full.view(2 * V, D)
    ss = (Sr * Sr).sum(1).double()
    ts_ = (T_full.view(2 * V, D) * Sr).sum(1).double()
    tt = (T_full.view(2 * V, D) ** 2).sum(1).double()
    qq = Q_full.view(2 * V, D).sum(1).double()
    b = torch.floor(torch.log2(n.clamp_min(2.0 ** -8))).clamp(-8, 24).long() + 8
    out = {}
    for tab, m in (("u", ~is_v), ("v", is_v)):
        rows = []
        for bi in range(33):
            sel = m & (b == bi) & (ss > 0)
            cnt = int(sel.sum())
            if cnt == 0:
                continue
            SS, TS, TT, QQ = float(ss[sel].sum()), float(ts_[sel].sum()), float(tt[sel].sum()), float(qq[sel].sum())
            rows.append({"log2n": bi - 8, "rows": cnt, "k": TS / SS, "explained": TS * TS / (SS * TT) if TT > 0 else 0.0,
                         "alignment": SS / QQ if QQ > 0 else 0.0, "share_of_S2": SS})
        tot = sum(r["share_of_S2"] for r in rows) or 1.0
        for r in rows:
            r["share_of_S2"] /= tot
        out[tab] = rows
    return out


rule = a.apply.split(":")
if rule[0] == "curve":
    curve_spec = json.load(open(rule[1]))
t0 = time.time()
launches = exchanges = since = 0
while True:
    for t in ts:
        t.train_step(a.positions)
    truth.train_step(a.positions)
    launches += 1; since += 1
    done = all(t.epoch_poll(0)[0] for t in ts)
    if not (done or since >= a.every):
        continue
    words = since * a.positions * per
    exchanges += 1
    T_full = truth.model_tensor() - base
    begun = [t.exchange_begin() for t in ts]
    if rule[0] == "smooth":
        k_rows = smooth_rows(words, float(rule[1]), float(rule[2]))
    elif rule[0] == "curve":
        k_rows = curve_rows(words, curve_spec)
    for c in range(begun[0][0]):
        bufs = [t.device_tensor(*t.exchange_delta(c)) for t in ts]
        st = torch.stack(bufs)
        s = st.sum(0)
        off = c * XCHUNK
        S_full[off:off + s.numel()] = s
        Q_full[off:off + s.numel()] = (st * st).sum(0)
        del st
        comb = T_full[off:off + s.numel()] if rule[0] == "oracle" else s * chunk_rows(k_rows, off, s.numel())
        for b_ in bufs:
            b_.copy_(comb)
        torch.cuda.synchronize()
        for t in ts:
            t.exchange_apply(c, 1.0)
    gw = sum(b_[1] for b_ in begun)
    for t in ts:
        t.exchange_end(gw)
    if exchanges in log_at:
        cv = pooled(words, T_full)
        res["curves"].append({"exchange": exchanges, "words_per_replica": words, "curves": cv})
        for tab in ("u", "v"):
            print("RT exchange %d (%d words/replica) table %s: " % (exchanges, words, tab) +
                  " ".join("n=2^%d:k=%.3f(e%.2f,a%.1f,s%.2f)" % (r["log2n"], r["k"], r["explained"], r["alignment"], r["share_of_S2"]) for r in cv[tab]), flush=True)
    # the next interval starts from the common model: the truth trainer follows it
    for t in ts:
        t.synchronize()
    if rule[0] != "oracle":
        truth.model_tensor().copy_(ts[0].model_tensor())
    base.copy_(truth.model_tensor())
    since = 0
    if done or (a.max_exchanges and exchanges >= a.max_exchanges):
        break
loss_r = sum(t.epoch_status()[3] for t in ts)
loss_t = truth.epoch_status()[3]
print("RT %d replicas, %d positions x %d launches per interval, apply %s: replicas' loss %.0f, truth trainer's loss %.0f (%+.2f %%), %d exchanges  [%.0f s]" % (
    R, a.positions, a.every, a.apply, loss_r, loss_t, 100 * (loss_r - loss_t) / abs(loss_t), exchanges, time.time() - t0), flush=True)
res.update({"replicas_loss": loss_r, "truth_loss": loss_t, "deviation_pct": 100 * (loss_r - loss_t) / abs(loss_t), "exchanges": exchanges})
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
for t in ts + [truth]:
    t.close()
corpus.close()
