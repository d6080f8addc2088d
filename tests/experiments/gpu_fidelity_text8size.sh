export TMPDIR=/tmp
python - <<'PY'
import numpy as np, time
rng = np.random.default_rng(0)
V, n = 70000, 17_000_000
w = 1.0 / np.arange(1, V, dtype=np.float64); cdf = np.cumsum(w); cdf /= cdf[-1]
ids = np.searchsorted(cdf, rng.random(n)) + 1
toks = np.char.add("w", ids.astype(str))
with open("/tmp/synth17m.txt", "w") as f:
    for i in range(0, n, 1000):
        f.write(" ".join(toks[i:i+1000])); f.write("\n")
PY
for k in 1 2; do for th in 64 0; do
  echo "== kernel $k (1 plain, 2 resident) -threads $th"
  W2B_WORKER_KERNEL=$k ./word2bits -train /tmp/synth17m.txt -output /tmp/o.bin -bitlevel 1 -size 200 -window 8 -negative 24 -threads $th -iter 3 -min-count 5 -binary 1 2>&1 | tr '\r' '\n' | grep -E "Hogwild|Epoch Loss" | tr '\n' ' '; echo
done; done
echo "== reference CPU 1 epoch -threads 64"
oracle/_ref/word2bits_stock -train /tmp/synth17m.txt -output /tmp/o_ref.bin -bitlevel 1 -size 200 -window 8 -negative 24 -threads 64 -iter 1 -min-count 5 -binary 1 2>&1 | tr '\r' '\n' | grep -E "Epoch Loss"
