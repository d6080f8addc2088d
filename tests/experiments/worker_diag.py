"""Diagnostics (GPU box): drift of the HIP worker form from the bit-reference oracle, next to the
drift between two builds of the oracle itself (FMA contraction on/off)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import word2bits_amd as w2b
from w2b_testlib import OracleState, zipf_ids


def stats(a, b):
    d = np.abs(a - b)
    return "sign %.4f mean %.3g med %.3g max %.3g" % (np.mean(np.signbit(a) == np.signbit(b)), d.mean(),
                                                    np.median(d), d.max())


def run(V, n, D, W, K, bl, sample, epochs, line=37):
    rng = np.random.default_rng(11)
    ids = zipf_ids(rng, V, n).astype(np.int32)
    ids[line::line] = 0
    cn = np.bincount(ids, minlength=V).astype(np.int64); cn[cn == 0] = 1
    tw = int(cn.sum())
    os_ = [OracleState(cn, D, window=W, negative=K, bitlevel=bl, num_threads=1, iters=epochs, sample=sample,
                       table_size=50000, fma=f) for f in (False, True)]
    t = w2b.Trainer(V, D, W, K, bl, num_threads=1, iter=epochs, sample=sample, train_words=tw)
    t.set_model(os_[0].u, os_[0].v)
    t.set_vocab_counts(cn, 50000); t.set_corpus(ids); t.set_shards(np.zeros(1, np.int64))
    for ep in range(epochs):
        lo = [o.train_epoch_tokens(ids, np.zeros(1, np.int64)) for o in os_]
        lg = t.train_epoch(777)
        fin, wca, alpha, _ = t.epoch_status()
        u, v = t.get_model()
        print("V=%d n=%d D=%d W=%d K=%d bl=%d s=%g ep%d | wca %s alpha %s | loss gpu %.3f ora %.3f fma %.3f" %
              (V, n, D, W, K, bl, sample, ep, wca == os_[0].m.word_count_actual,
               np.float32(alpha) == np.float32(os_[0].m.alpha), lg, lo[0], lo[1]))
        print("    u gpu-ora: %s | fma-ora: %s" % (stats(u, os_[0].u), stats(os_[1].u, os_[0].u)))
        print("    v gpu-ora: %s | fma-ora: %s" % (stats(v, os_[0].v), stats(os_[1].v, os_[0].v)))
    t.close()


for cfg in [(150, 30000, 64, 5, 5, 1, 1e-3, 2), (150, 30000, 100, 8, 24, 2, 0.0, 2), (150, 30000, 800, 8, 24, 1, 0.0, 2),
            (5000, 3000, 64, 5, 5, 1, 0.0, 1), (5000, 3000, 200, 8, 24, 1, 0.0, 1), (5000, 3000, 200, 8, 24, 2, 0.0, 1),
            (5000, 3000, 200, 8, 24, 0, 0.0, 1), (20000, 3000, 800, 8, 24, 1, 1e-3, 1)]:
    run(*cfg)
