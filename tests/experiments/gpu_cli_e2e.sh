#!/bin/bash
# end-to-end timing of the ./word2bits command line on a text8-sized synthetic corpus (17M tokens, ~70K words)
set +e
export TMPDIR=/tmp
python - <<'PY'
import numpy as np, time
rng = np.random.default_rng(0)
V, n = 70000, 17_000_000
w = 1.0 / np.arange(1, V, dtype=np.float64); cdf = np.cumsum(w); cdf /= cdf[-1]
ids = np.searchsorted(cdf, rng.random(n)) + 1
toks = np.char.add("w", ids.astype(str))
t0 = time.time()
with open("/tmp/synth17m.txt", "w") as f:
    f.write(" ".join(toks))        # one line, like text8
print("corpus written", time.time() - t0)
PY
ls -la /tmp/synth17m.txt
for th in 0 12; do
  echo "== ./word2bits -threads $th (bitlevel 1, size 200, window 8, negative 24, iter 5)"
  S=$(date +%s.%N)
  ./word2bits -train /tmp/synth17m.txt -output /tmp/o_$th.bin -bitlevel 1 -size 200 -window 8 -negative 24 -threads $th -iter 5 -min-count 5 -binary 1 -debug 1 2>&1 | tr '\r' '\n' | grep -E "Vocab|Words in|Hogwild|Epoch Loss|word2bits:" | head -12
  E=$(date +%s.%N); echo "wall seconds: $(python -c "print(round($E-$S,2))")"
done
echo "== reference CPU, 1 epoch, -threads 64 (for scale)"
S=$(date +%s.%N)
oracle/_ref/word2bits_stock -train /tmp/synth17m.txt -output /tmp/o_ref.bin -bitlevel 1 -size 200 -window 8 -negative 24 -threads 64 -iter 1 -min-count 5 -binary 1 -debug 1 2>&1 | grep -E "Epoch Loss"
E=$(date +%s.%N); echo "wall seconds: $(python -c "print(round($E-$S,2))")"
