"""diagnostic, not a test: first divergence of the row-group kernel from the plain kernel, one worker, growing launch counts"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import word2bits_amd as w2b
from test_gpu_worker import token_stream, counts_of

D = int(sys.argv[1]) if len(sys.argv) > 1 else 200
K = int(sys.argv[2]) if len(sys.argv) > 2 else 24
kn = {}
for a in sys.argv[3:]:
    k, v = a.split("="); kn[k] = int(v)
V, n = 300, 6000
rng = np.random.default_rng(9)
ids = token_stream(rng, V, n, line=23)
cn = counts_of(ids, V)

def run(groups, npos, per=1000):
    t = w2b.Trainer(V, D, 8, K, 1, num_threads=1, iter=1, sample=0.0, train_words=int(cn.sum()), compute_loss=True, row_groups=groups, **kn)
    t.init_net(); t.set_vocab_counts(cn, 50000); t.set_corpus(ids); t.set_shards(np.zeros(1, np.int64))
    t.epoch_begin()
    left = npos
    while left > 0:
        t.train_step(min(per, left)); left -= min(per, left)
    fin, wca, alpha, loss = t.epoch_status()
    u, v = t.get_model(); name = t.worker_kernel_name(); t.close()
    return name, u, v, wca, loss

for npos in (1, 2, 3, 4, 6, 10, 30, 100, 1000):
    a = run(False, npos); b = run(True, npos)
    du = np.where((a[1].view(np.uint32) != b[1].view(np.uint32)).any(1))[0]
    dv = np.where((a[2].view(np.uint32) != b[2].view(np.uint32)).any(1))[0]
    print(npos, a[0], b[0], "u rows differ:", du[:12].tolist(), len(du), "v rows differ:", dv[:12].tolist(), len(dv),
          "max|du| %.3g max|dv| %.3g" % (np.abs(a[1] - b[1]).max(), np.abs(a[2] - b[2]).max()), "loss", a[4], b[4], flush=True)
    if len(du) + len(dv) > 0 and npos >= 3:
        r = (du.tolist() + dv.tolist())[0]
        tab = 1 if len(du) else 2
        cols = np.where(a[tab][r].view(np.uint32) != b[tab][r].view(np.uint32))[0]
        print("  first differing row", r, "table", "uv"[tab - 1], "cols", cols[:16].tolist(), len(cols), "plain", a[tab][r][cols[:4]], "groups", b[tab][r][cols[:4]])
        print("  sentence start:", ids[:12].tolist())
        break
