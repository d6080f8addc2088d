"""Measurement, not a test: training effect of the replica exchange on ONE GPU (R replicas in this process, the collective
supplied through the phase API -- tests/test_gpu_exchange.py run_replicas) over launch length x exchange scheme.
  python tests/experiments/exchange_matrix.py [--positions 1024,256,64] [--replicas 2,4]
Schemes per (R, positions): none (end of epoch only) | full every launch (mode 2) | full every E launches.  (Round 4's two-tier
schemes -- a hot tier of B MB per table after every launch -- were removed with the hot tier in round 5.)
Printed: epoch loss and its deviation from the single replica at the same positions."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.init()        # torch's bundled HIP runtime finds the GPU only when it initialises before the library's (tests/conftest.py)
import word2bits_amd as w2b
from w2b_testlib import write_zipf_text_corpus
from test_gpu_exchange import run_replicas

ap = argparse.ArgumentParser()
ap.add_argument("--positions", default="1024,256,64")
ap.add_argument("--replicas", default="2,4")
ap.add_argument("--every", default="8,32", help="full exchange every E launches")
ap.add_argument("--sat", default="0", help="w2b_tuning.exchange_sat_updates values to sweep (0 = library default)")
ap.add_argument("--modes", default="2", help="exchange modes for the 'full every launch' scheme (0 delta-sum, 2 contributor mean)")
a = ap.parse_args()
path = write_zipf_text_corpus("/tmp/w2b_xm_t8.txt")
corpus = w2b.Corpus(path, 5)
flags = dict(bitlevel=1, size=200, window=8, negative=24)
probe = w2b.Trainer(2, 200, 8, 24, 1, num_threads=1, train_words=corpus.train_words)
workers = probe.suggested_threads(); probe.close()
workers -= workers % 4
for positions in [int(x) for x in a.positions.split(",")]:
    t0 = time.time()
    one, launches = run_replicas(corpus, 1, workers, 1, positions, flags)
    print("XM positions=%d workers=%d launches/epoch=%d: 1 replica loss %.0f  [%.1f s]" % (positions, workers, launches, one, time.time() - t0), flush=True)
    for R in [int(x) for x in a.replicas.split(",")]:
        schemes = [("none", dict(sync_every=0)), ("full every launch", dict(sync_every=1))]
        for E in [int(x) for x in a.every.split(",")]:
            schemes.append(("full every %d only" % E, dict(sync_every=E)))
        seen = set()
        for sat in [int(x) for x in a.sat.split(",")]:
            for name, kw in schemes:
                for mode in ([int(x) for x in a.modes.split(",")] if name == "full every launch" else [2]):
                    key = (name, mode, sat if name != "none" else 0)
                    if key in seen:
                        continue
                    seen.add(key)
                    t0 = time.time()
                    tn = dict(exchange_sat_updates=sat) if sat else {}
                    try:
                        loss, _ = run_replicas(corpus, R, workers, positions=positions, flags=flags, mode=mode, **kw, **tn)
                    except Exception as e:
                        print("XM positions=%-5d R=%d %-26s mode %d sat %-5d FAILED %r" % (positions, R, name, mode, sat, e), flush=True)
                        continue
                    print("XM positions=%-5d R=%d %-26s mode %d sat %-5d loss %.0f (%+.2f %% vs 1 replica)  [%.1f s]" % (
                        positions, R, name, mode, sat, loss, 100 * (loss - one) / abs(one), time.time() - t0), flush=True)
corpus.close(); os.remove(path)
