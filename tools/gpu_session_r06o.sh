#!/bin/bash
# round 6, session o: the exchange at bitlevel 0 (no quantization cells: mode 2 is the saturation factor alone) -- 8 replicas x 128
# workers on the 22 M-token proxy at the configs[1] shape, full precision, at the automatic interval and at twice / half of it
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r06o
mkdir -p $OUT
python - <<'PY'
import sys
sys.path.insert(0, "tests")
from w2b_testlib import write_headline_corpus
write_headline_corpus("/tmp/headline.txt")
PY
RR="timeout 1500 python tests/experiments/replica_rules.py"
for pos in 672 336 1344; do
  $RR /tmp/headline.txt --bitlevel 0 --positions $pos --rules "lib2" --out $OUT/rules_b0_p$pos.json 2>&1 | grep RR | tee -a $OUT/rules_b0.txt
done
$RR /tmp/headline.txt --bitlevel 2 --positions 672 --rules "lib2" --out $OUT/rules_b2_p672.json 2>&1 | grep RR | tee -a $OUT/rules_b2.txt
rm -f /tmp/headline.txt
echo "== done"
