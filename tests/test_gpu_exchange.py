"""-m gpu: the replica exchange of the library (w2b_sync_replicas / w2b_exchange_*), as far as ONE GPU allows.

RCCL does not put two ranks on one device, so the N > 1 collective itself cannot run on this pool.  Everything around
it can:
  * a communicator of size 1 (w2b_comm_init with an id) drives the whole RCCL path -- base snapshot, delta kernel,
    ncclAllReduce, apply kernel, the progress counters, two exchange streams, fences -- and must leave training
    bit-identical to a trainer without a communicator;
  * the phase API (w2b_exchange_begin / delta / apply / end) is the same code with the sum supplied by the host: R
    replicas live in this process on one GPU and the "collective" is a torch sum of their delta buffers.  That pins the
    arithmetic (W_r += sum - d_r, base += sum) against host copies, and -- the point of this file -- the TRAINING EFFECT of
    exchanging only every k launches: 2 and 4 replicas against one replica with the same total number of workers on the
    text8-sized corpus (ref src/word2bits.cpp:535-536 runs all threads on one shared model; delta-sum adds every
    replica's update of a hot row on top of the others', which is where a too lazy exchange would show)."""
import json
import os

import numpy as np
import pytest

import word2bits_amd as w2b
from word2bits_amd import replicas
from w2b_testlib import GOLDEN

pytestmark = pytest.mark.gpu


def small_counts(seed=3, V=3000, n=60000):
    rng = np.random.default_rng(seed)
    ids = (rng.zipf(1.3, n) % (V - 1) + 1).astype(np.int32)
    ids[49::50] = 0
    return ids, np.maximum(np.bincount(ids, minlength=V), 1).astype(np.int64)


def small_setup(nw, total, offset, seed=3, V=3000, D=64, n=60000, bitlevel=1, **kw):
    ids, counts = small_counts(seed, V, n)
    t = w2b.Trainer(V, D, 5, 5, bitlevel, num_threads=nw, iter=1, sample=0.0, train_words=int(counts.sum()),
                    compute_loss=True, worker_offset=offset, total_threads=total, **kw)
    t.init_net()
    t.set_vocab_counts(counts, 100000)
    t.set_corpus(ids)
    t.set_shards(replicas.token_shard_starts(len(ids), total, offset, nw))
    return t


def flat(t):
    return np.concatenate([x.ravel() for x in t.get_model()])


def saturated_rows(counts, words, window, negative):
    """the library's rule for mode 2 restated (w2b_trainer.cpp xchg_saturated): per table, rows 1..n with n the result of
    the same binary search over `rate x words >= 32` (the vocabulary is meant to be sorted by count)"""
    c = counts.astype(np.float64)
    pw, tot = (c ** 0.75).sum(), c.sum()
    rate = {"v": negative * c ** 0.75 / pw + c / tot, "u": (window + 1) * c / tot}
    V = len(c)
    out = np.zeros(2 * V, bool)
    for tab, off in (("u", 0), ("v", V)):
        lo, hi = 0, V - 1
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if rate[tab][mid] * words >= 32.0:
                lo = mid
            else:
                hi = mid - 1
        out[off + 1:off + lo + 1] = True
    return out


def test_size_one_communicator_runs_the_whole_exchange_and_changes_nothing(gpu):
    """the RCCL exchange path had never executed.  One rank, a real communicator: every exchange must
    leave the model bit-identical, the alpha schedule untouched (wca_others = 0), and the run equal to a trainer that
    never exchanges -- deterministic here because there is one worker."""
    res = []
    for with_comm in (False, True):
        t = small_setup(1, 1, 0)
        if with_comm:
            t.comm_init(1, 0, w2b.comm_unique_id())
        t.epoch_begin()
        for k in range(12):
            t.train_step(300)
            if with_comm and k % 3 == 2:
                before = None
                if k == 5:
                    t.synchronize()
                    before = flat(t)
                t.sync_replicas((k // 3) % 3)                # all three modes
                if before is not None:
                    assert np.array_equal(before.view(np.uint32), flat(t).view(np.uint32))
        fin, wca, alpha, loss = t.epoch_status()
        res.append((flat(t), wca, alpha, loss))
        if with_comm:
            n, ms = t.sync_stats()
            assert n == 4 and ms > 0
        t.close()
    assert np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
    assert res[0][1:] == res[1][1:]


def local_exchange(ts, mode=0, counts_out=None):
    """the host-supplied collective of the phase API for replicas that live in this process: sum of the delta buffers
    (counts_out: a list that receives the summed per-row contributor counts of mode 2 as a numpy array)"""
    import torch
    begun = [t.exchange_begin() for t in ts]
    n_chunks = begun[0][0]
    scale = 1.0 / len(ts) if mode == 1 else 1.0
    if mode == 2:                                     # contributor counts: how many replicas changed each row
        cnts = [t.device_tensor(*t.exchange_counts()) for t in ts]
        total = torch.stack(cnts).sum(0)
        if counts_out is not None:
            counts_out.append(total.cpu().numpy())
        for b in cnts:
            b.copy_(total)
        torch.cuda.synchronize()
    for c in range(n_chunks):
        bufs = [t.device_tensor(*t.exchange_delta(c)) for t in ts]
        total = torch.stack(bufs).sum(0)
        for b in bufs:
            b.copy_(total)
        torch.cuda.synchronize()
        for t in ts:
            t.exchange_apply(c, scale)
    words = sum(b[1] for b in begun)
    for t in ts:
        t.exchange_end(words)
    return words


def smooth_factors(counts, words, window, negative, c, tau_u=64.0, tau_v=64.0):
    """the library's default combination rule of mode 2 restated (w2b_kernels_misc.hip k_xchg_factor, w2b_trainer.cpp
    xchg_upload_rates; -sample 0): per row of [u || v], with n = expected updates per replica since the last exchange and c =
    replicas that changed the row,  k = (1 - exp(-c n / tau)) / (c (1 - exp(-n / tau)))  for c > 1, else 1"""
    cn = counts.astype(np.float64)
    V = len(cn)
    kept_tot, pw = cn[1:].sum(), (cn ** 0.75).sum()
    rate = np.zeros(2 * V)
    rate[1:V] = (window + 1) * cn[1:] / kept_tot
    rate[V + 1:] = negative * cn[1:] ** 0.75 / pw + cn[1:] / kept_tot
    tau = np.concatenate([np.full(V, tau_u), np.full(V, tau_v)])
    x = np.float32(rate).astype(np.float64) * np.float32(words) / tau
    k = np.ones(2 * V)
    m = (c > 1) & (x > 1e-6)
    k[m] = np.expm1(-c[m] * x[m]) / (c[m] * np.expm1(-x[m]))
    return np.clip(k, 1.0 / np.maximum(c, 1), 1.0)


def host_cell(x, bitlevel):
    """the quantization cell of every element (quantize(), ref :73-108), as labels: equal labels <=> equal forward values"""
    neg = (x < 0).astype(np.int64)
    if bitlevel == 1:
        return neg
    assert bitlevel == 2
    return neg * 2 + (~(np.abs(x) <= np.float32(0.5))).astype(np.int64)


@pytest.mark.parametrize("mode", [0, 1, 2, "2 two bits", "2 saturation only", "2 hard threshold"])
def test_phase_api_arithmetic_two_replicas(gpu, mode):
    """W_r += comb - d_r on top of the CURRENT rows, base += comb, against host arithmetic on copies of both replicas.
    comb = a * sum with a = 1 (mode 0) / 1/R (mode 1); mode 2: a per-row factor k on the sum -- exponential saturation
    (exchange_rule 2), or rounds 4-5's 1 / contributors for the saturated rows (exchange_rule 1) -- and, by default (exchange_rule
    0) at ONE bit, per element the whole sum wherever base + sum has the sign of base + k * sum (its quantization cell); at two bits
    exchange_rule 0 is the saturation factor alone (the cells were measured to do harm there: -9.4 % against -1.9 %)."""
    R, nw = 2, 4
    rule = {"2 saturation only": 2, "2 hard threshold": 1}.get(mode, 0)
    bitlevel = 2 if mode == "2 two bits" else 1
    if isinstance(mode, str):
        mode = 2
    ts = [small_setup(nw, R * nw, r * nw, seed=3, bitlevel=bitlevel, **(dict(exchange_rule=rule) if rule else {})) for r in range(R)]
    for t in ts:
        t.exchange_init()
        t.epoch_begin()
    base = flat(ts[0])
    assert np.array_equal(base, flat(ts[1]))
    V2, D = 2 * 3000, 64
    for rnd in range(3):
        for t in ts:
            for _ in range(3):
                t.train_step(150)
        mine = [flat(t) for t in ts]
        d = [m - base for m in mine]
        S = d[0] + d[1]
        # (the exchange runs first: mode 2's per-row contributor counts are the DEVICE's -- "this replica's row differs from base, bit
        # for bit" -- the one input of the rule that host copies cannot reproduce: host and device bases agree to rounding only)
        counts = []
        words = local_exchange(ts, mode, counts)
        got = [flat(t) for t in ts]
        alt = None                                                          # the other admissible value where a sign is decided by rounding
        if mode == 2:
            c = counts[0]
            assert c.max() == 2 and (c >= (np.abs(S).reshape(V2, D).max(1) > 1e-5)).all()
            nwords = 3 * 150 * nw                                           # 3 launches x 150 positions x nw workers
            if rule == 1:
                sat = saturated_rows(small_counts(3)[1], nwords, 5, 5)
                assert sat.any() and not sat.all() and c[sat].max() == 2
                k = np.where(sat, np.float32(1) / np.maximum(c, 1), np.float32(1))
            else:
                k = smooth_factors(small_counts(3)[1], nwords, 5, 5, c)
                both = c == 2
                assert both.any() and k[both].min() < 0.55 and k[both].max() > 0.95   # from the mean to the sum, and in between
                assert ((k[both] > 0.6) & (k[both] < 0.9)).any()
                assert (k[~both] == 1).all()
            safe = k.astype(np.float32)[:, None].repeat(D, 1).ravel() * S
            if rule == 0 and bitlevel == 1:
                same = host_cell(base + safe, bitlevel) == host_cell(base + S, bitlevel)
                assert 0.5 < same.mean() < 1.0 and (~same).sum() > 100              # both branches are exercised
                total, alt = np.where(same, S, safe), np.where(same, safe, S)
            else:
                total = safe
        else:
            total = np.float32(1.0 if mode == 0 else 1.0 / R) * S
        tol = 2e-6 + 1e-5 * np.abs(S).max() * (mode == 2 and rule != 1)
        for r in range(R):
            err = np.abs(got[r] - (mine[r] + (total - d[r])))
            if alt is not None:                                             # (an element whose sign hangs on the last bit may take the other branch)
                bad = err > tol
                assert bad.mean() < 1e-4, (rnd, r, bad.mean())
                err = np.where(bad, np.abs(got[r] - (mine[r] + (alt - d[r]))), err)
            assert err.max() <= tol, (rnd, r, err.max())
            assert np.abs(got[r] - mine[r]).max() > 0            # the other replica's work arrived
        # the common state after the exchange: exact host arithmetic where comb is exact on the host; with the saturation factors
        # (float expm1 on the device) the device's own result -- no training since the deltas, so got = base + comb up to rounding
        base = got[0].copy() if (mode == 2 and rule != 1) else base + total
        assert words == sum(t.epoch_status(want_loss=False)[1] for t in ts)
    assert np.abs(got[0] - got[1]).max() <= 4e-6                 # the replicas agree after every exchange
    for t in ts:
        t.close()


def test_exchange_needs_init(gpu):
    t = small_setup(2, 2, 0)
    with pytest.raises(w2b.W2bError) as e:
        t.exchange_begin()
    assert "exchange_init" in str(e.value)
    t.close()


# ---------------------------------------------------------------------------------------------- training effect
def run_replicas(corpus, R, workers_total, sync_every, positions, flags, slices=True, mode=2, **tuning):
    """one epoch over `corpus` with R replicas of workers_total / R workers each, exchanged every sync_every launches
    and at the end (sync_every 0: at the end only), as ./word2bits -gpus N does; returns the summed epoch loss"""
    per = workers_total // R
    starts, ov = corpus.shards(workers_total)
    tokens = corpus.tokens()
    quota = corpus.train_words // workers_total
    ts = []
    for r in range(R):
        t = w2b.Trainer(corpus.vocab_size, flags["size"], flags["window"], flags["negative"], flags["bitlevel"],
                        num_threads=per, iter=1, train_words=corpus.train_words, compute_loss=True,
                        worker_offset=r * per, total_threads=workers_total, **tuning)
        t.init_net()
        t.set_vocab_counts(corpus.counts(), 100_000_000)
        st = starts[r * per:(r + 1) * per]
        if slices and R > 1:
            lo, hi, more = replicas.replica_token_slice(tokens, st, quota)
            t.set_corpus_slice(tokens[lo:hi], more)
            t.set_shards(st - lo, ov[r * per:(r + 1) * per])
        else:
            t.set_corpus(tokens)
            t.set_shards(st, ov[r * per:(r + 1) * per])
        if R > 1:
            t.exchange_init()
        t.epoch_begin()
        ts.append(t)
    launches = 0
    while True:
        for t in ts:
            t.train_step(positions)
        launches += 1
        done = all(t.epoch_poll(0)[0] for t in ts)
        if R > 1 and (done or (sync_every > 0 and launches % sync_every == 0)):
            local_exchange(ts, mode)
        if done:
            break
    loss = sum(t.epoch_status()[3] for t in ts)
    models = [flat(t) for t in ts] if R > 1 else None
    for t in ts:
        t.close()
    if models:                                  # after the final exchange every replica holds the same model
        for m in models[1:]:
            assert np.abs(m - models[0]).max() <= 1e-5 * max(1.0, float(np.abs(models[0]).max()))
    return loss, launches


def test_training_effect_of_the_exchange_text8_size(gpu, tmp_path_factory):
    """17 M tokens, 70 K words, bitlevel 1, size 200, window 8, negative 24, one epoch.  The number of workers is what
    `./word2bits -threads 0` picks for the file; it is split over 1, 2 and 4 replicas.

    Round 3 exchanged once per launch of 1024 positions and found the exchange worth no more than not exchanging
    (-2.5 % / -6.8 % against -2.6 % / -7.2 %).  Round 4 (profiles/r04_sessions/r04d_exchange_matrix.txt, r04e, r04f): what
    matters is the INTERVAL -- with launches of 256 positions (22-44 K centre words per replica between two exchanges) the
    contributor-mean exchange (mode 2) after every launch keeps 2 replicas within 0.7-0.8 % and 4 within 1.4-1.6 % of the
    single replica, 2 and 6 points better than no exchange.  Asserted: within EXCHANGE_RTOL and at least 1 point better than
    exchanging at the end of the epoch only.  Printed for the record: the plain delta-sum (mode 0: over-shoots, -3 % / -9 %),
    and a full exchange every 8 launches only.  (Round 4 also had a hot tier -- the leading rows only, after every other
    launch: measured no better than without it, the rows that are rare individually are 28 % of all negative draws and 12 % of
    all context positions and want the short interval as much as the frequent ones; removed in round 5.)"""
    from w2b_testlib import write_zipf_text_corpus
    d = tmp_path_factory.mktemp("xchg")
    path = write_zipf_text_corpus(str(d / "c.txt"))
    corpus = w2b.Corpus(path, 5)
    flags = dict(bitlevel=1, size=200, window=8, negative=24)
    probe = w2b.Trainer(2, 200, 8, 24, 1, num_threads=1, train_words=corpus.train_words)
    workers = probe.suggested_threads()
    probe.close()
    workers -= workers % 4
    positions = 256
    one, launches = run_replicas(corpus, 1, workers, 1, positions, flags)
    print("EXCHANGE text8size workers=%d launches/epoch=%d: 1 replica loss %.0f" % (workers, launches, one))
    dev = {}
    for R in (2, 4):
        for name, kw in (("none", dict(sync_every=0)), ("every launch, mode 2", dict(sync_every=1)),
                         ("every launch, mode 0", dict(sync_every=1, mode=0)), ("every 8 launches", dict(sync_every=8))):
            if R == 2 and name.startswith("every 8"):
                continue
            loss, _ = run_replicas(corpus, R, workers, positions=positions, flags=flags, **kw)
            dev[(R, name)] = (loss - one) / abs(one)
            print("EXCHANGE text8size replicas=%d %-26s loss %.0f (%+.2f %% vs 1 replica)" % (R, name, loss, 100 * dev[(R, name)]))
    for R in (2, 4):
        got, none = dev[(R, "every launch, mode 2")], dev[(R, "none")]
        assert abs(got) <= EXCHANGE_RTOL[R], (R, got)
        assert abs(none) - abs(got) >= 0.01, (R, got, none)           # at least one point better than not exchanging
    corpus.close()


# Epoch-loss tolerance against the single replica, by number of replicas, for the contributor-mean exchange after every
# launch of 256 positions (measured: 2 replicas -0.66 ... -0.82 %, 4 replicas -1.39 ... -1.59 %; round 3's tolerances for
# one exchange per 1024 positions were 5 % / 10 %).
EXCHANGE_RTOL = {2: 0.015, 4: 0.03}


EIGHT_REPLICAS_RTOL = 0.04


def test_eight_replicas_at_the_configs3_shape(gpu, tmp_path):
    """BASELINE configs[3] on ONE GPU through the phase API: 8 replicas at the configs[1] shape (V = 400 K, size 800, negative
    24, bitlevel 1), 128 workers each, on the 22 M-token proxy file, against the single replica with the same 1024 workers and the
    same launches; a full exchange after every launch of 672 positions = 86 K centre words per replica (what `./word2bits -gpus 8`
    picks for this file: 1 / 32 of a replica's epoch).  Round 5 RECORDED this at -9 % (mean of the contributors for saturated rows)
    and called the path "built, not faithful".  Round 6 (profiles/r06_sessions/), at 131 K words: the
    exponential saturation factor alone -7.6 %; the per-row least-squares factor measured against a truth run -18 % (diverges in
    closed loop); the shipped rule -- saturation decides every element's quantized value, the whole sum is taken wherever it stays
    in that quantization cell -- -2.6 ... -2.9 % (three repeats within 0.06 points), +0.8 % at 65 K words, -8 % at 262 K, where
    a PERFECT rule (every replica adopts the single replica's model at every exchange) ends -4.9 % at 131 K: what is left is the
    interval itself, not the rule.  Gate: within EIGHT_REPLICAS_RTOL of the single replica, and >= 15 points better than meeting
    at the end of the epoch only (-24 %).  The literal 100 M-token stream: test_eight_replicas_literal_stream."""
    from w2b_testlib import write_headline_corpus
    path = write_headline_corpus(str(tmp_path / "c.txt"))
    corpus = w2b.Corpus(path, 5)
    flags = dict(bitlevel=1, size=800, window=8, negative=24)
    try:
        positions = 672                                            # 86 K centre words per replica and launch
        one, launches = run_replicas(corpus, 1, 1024, 1, positions, flags, sample=0.0)
        none, _ = run_replicas(corpus, 8, 1024, 0, positions, flags, sample=0.0)
        every, _ = run_replicas(corpus, 8, 1024, 1, positions, flags, sample=0.0)
        d_none, d_every = (none - one) / abs(one), (every - one) / abs(one)
        print("EXCHANGE configs[3] shape, 8 replicas x 128 workers, %d launches: 1 replica %.0f | end of epoch only %+.2f %% | "
              "after every launch of 86 K words %+.2f %%" % (launches, one, 100 * d_none, 100 * d_every))
        assert -0.35 <= d_none <= -0.15
        assert abs(d_every) <= EIGHT_REPLICAS_RTOL, d_every
        assert d_every - d_none >= 0.15
    finally:
        corpus.close()
        os.remove(path)


def test_eight_replicas_literal_stream(gpu, tmp_path):
    """The same 8 replicas on BASELINE configs[1] LITERALLY per job (100 M tokens; 12.5 M words per replica), a full exchange after
    every launch of 3072 positions = 393 K centre words per replica (the automatic interval: 1 / 32 of a replica's epoch) and of
    8192 positions = 1 M words -- the interval at which one exchange of the whole 2.56 GB model per launch fits the xGMI links (a
    launch of 1 M words is 38 ms on a full device; ring over one link 29 ms, reduce-scatter + all-gather over all seven 4 ms).
    Measured in round 6: +5.0 / +2.0 / +0.2 / -2.1 % of the single replica's epoch loss at 131 K / 262 K / 524 K / 1 M words (round
    5's rule at 1 M words: -12.6 %, worse than not exchanging at all).  Gate: 3 % each.  (The single replica is run with the same launches: its own epoch loss moves by 1.6 %
    between launches of 1024 and 8192 positions -- a launch boundary is a device-wide barrier.)"""
    from w2b_testlib import write_headline_corpus
    path = write_headline_corpus(str(tmp_path / "c.txt"), n_zipf=98_000_000)
    corpus = w2b.Corpus(path, 5)
    flags = dict(bitlevel=1, size=800, window=8, negative=24)
    try:
        for positions in (3072, 8192):
            one, launches = run_replicas(corpus, 1, 1024, 1, positions, flags, sample=0.0)
            every, _ = run_replicas(corpus, 8, 1024, 1, positions, flags, sample=0.0)
            d_every = (every - one) / abs(one)
            print("EXCHANGE configs[1] literally, 8 replicas x 128 workers, %d launches of %d K words per replica: 1 replica %.0f | "
                  "8 replicas %+.2f %%" % (launches, positions * 128 // 1000, one, 100 * d_every))
            assert abs(d_every) <= 0.03, (positions, d_every)
    finally:
        corpus.close()
        os.remove(path)
