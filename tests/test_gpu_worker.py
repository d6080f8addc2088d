"""-m gpu: form (i) -- the on-device restatement of TrainModelThread (sentence reader, sub-sampling,
window draw, negative draws, alpha schedule) against the CPU oracle on the same token stream."""
import numpy as np
import pytest

import word2bits_amd as w2b
from w2b_testlib import OracleState, zipf_ids

pytestmark = pytest.mark.gpu


def token_stream(rng, V, n, line=37):
    ids = zipf_ids(rng, V, n).astype(np.int32)
    ids[line::line] = 0                      # "</s>" every `line` tokens
    ids[5] = 0
    ids[6] = 0                               # an empty sentence (ref :428 still draws)
    return ids


def counts_of(ids, V):
    cn = np.bincount(ids, minlength=V).astype(np.int64)
    cn[cn == 0] = 1
    return cn


@pytest.mark.parametrize("bitlevel,sample,D,window,negative", [
    (1, 1e-3, 64, 5, 5),
    (0, 1e-2, 48, 3, 7),
    (2, 0.0, 100, 8, 24),
    (1, 0.0, 800, 8, 24),
])
def test_single_worker_matches_oracle(gpu, bitlevel, sample, D, window, negative):
    V, n = 150, 30000
    rng = np.random.default_rng(11)
    ids = token_stream(rng, V, n)
    ids[2000:3300] = zipf_ids(rng, V, 1300)  # one 1300-token line: sentence chunking at 1000 (ref :410)
    cn = counts_of(ids, V)
    tw = int(cn.sum())
    o = OracleState(cn, D, window=window, negative=negative, bitlevel=bitlevel, num_threads=1, iters=2,
                    sample=sample, table_size=50000)
    t = w2b.Trainer(V, D, window, negative, bitlevel, num_threads=1, iter=2, alpha=0.05, sample=sample,
                    train_words=tw, compute_loss=True)
    o.m.train_words = tw
    t.set_model(o.u, o.v)
    t.set_vocab_counts(cn, 50000)
    t.set_corpus(ids)
    t.set_shards(np.zeros(1, np.int64))
    for ep in range(2):
        lo = o.train_epoch_tokens(ids, np.zeros(1, np.int64))
        lg = t.train_epoch(positions_per_launch=777)     # odd launch size: exercises save/restore
        fin, wca, alpha, _ = t.epoch_status()
        assert fin
        assert wca == o.m.word_count_actual                      # integer bookkeeping: exact
        assert np.float32(alpha) == np.float32(o.m.alpha)        # alpha staircase: exact
        assert lg == pytest.approx(lo, rel=2e-3)
    u, v = t.get_model()
    du, dv = np.abs(u - o.u), np.abs(v - o.v)
    if bitlevel == 1:
        assert np.mean(np.signbit(u) == np.signbit(o.u)) >= 0.99
        assert np.median(du) <= 1e-4
    else:
        assert du.mean() <= 1e-4 and dv.mean() <= 1e-4, (du.mean(), dv.mean())
        assert du.max() <= 5e-2
    t.close()


def test_shard_override_and_multi_worker_bookkeeping(gpu):
    """4 workers (Hogwild): values are racy, but the integer bookkeeping is deterministic."""
    V, n, D = 120, 20000, 32
    rng = np.random.default_rng(2)
    ids = token_stream(rng, V, n)
    cn = counts_of(ids, V)
    tw = int(cn.sum())
    starts = np.array([0, 5000, 10001, 15002], np.int64)
    ov = np.array([-2, 7, -1, 0], np.int32)      # truncated first word: in vocab / OOV / "</s>"
    o = OracleState(cn, D, window=5, negative=5, bitlevel=1, num_threads=4, iters=1, sample=1e-3,
                    table_size=50000)
    o.m.train_words = tw
    t = w2b.Trainer(V, D, 5, 5, 1, num_threads=4, iter=1, sample=1e-3, train_words=tw)
    t.set_model(o.u, o.v)
    t.set_vocab_counts(cn, 50000)
    t.set_corpus(ids)
    t.set_shards(starts, ov)
    o.train_epoch_tokens(ids, starts, ov)
    t.train_epoch(positions_per_launch=500)
    fin, wca, alpha, loss = t.epoch_status()
    assert fin and wca == o.m.word_count_actual
    u, v = t.get_model()
    assert np.isfinite(u).all() and np.isfinite(v).all()
    assert np.abs(u - o.u).mean() < 5e-2
    t.close()
