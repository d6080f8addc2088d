#!/bin/bash
# round 6, session n: the copies are folded on a word schedule (every 4096 positions per worker, and at every reader of the model)
# instead of after every launch: the whole -m gpu suite, the driver's bench command, and the single replica on the literal stream at
# three launch lengths (round 6 measured -1.3 / -0.5 / +0.3 % of the reference at 1024 / 4096 / 8192 positions with per-launch folds)
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r06n
mkdir -p $OUT
python - <<'PY'
import sys
sys.path.insert(0, "tests")
from w2b_testlib import write_headline_corpus
write_headline_corpus("/tmp/cfg1_100m.txt", n_zipf=98_000_000)
PY
F="-bitlevel 1 -size 800 -window 8 -negative 24 -iter 1 -sample 0 -min-count 5 -binary 1 -threads 1024"
for pos in 1024 4096 8192 1024 8192; do
  ./word2bits -train /tmp/cfg1_100m.txt -output /dev/null $F -positions $pos > $OUT/run.txt 2> $OUT/run.err
  python - $pos <<PY | tee -a $OUT/launch_length.txt
import re, sys
out = open("$OUT/run.txt").read().replace("\r", "\n")
L = [float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", out)]
ref = -543668877.0
print("LAUNCH positions %s: loss %.0f deviation %+.2f %% of the reference's band" % (sys.argv[1], L[0], 100 * (L[0] - ref) / abs(ref)))
PY
done
rm -f /tmp/cfg1_100m.txt
timeout 2400 python -m pytest tests -q -m gpu -s 2>&1 | grep -E "FIDELITY|EXCHANGE|passed|failed|FAILED|Error" | cut -c1-300 | tee $OUT/pytest_gpu.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.log 2>/dev/null; tail -1 $OUT/bench_default.log | cut -c1-400
echo "== done"
