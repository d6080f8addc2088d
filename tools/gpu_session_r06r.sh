#!/bin/bash
# round 6, session r: at ONE bit, is the cell rule better than the saturation factor alone outside the configs[1] shape too?
# long_d200 (size 200, V = 70 K, default sub-sampling) and heldout_v1m (size 512, V = 1 M), 8 replicas, both rules
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r06r
mkdir -p $OUT
python - <<'PY'
import sys
sys.path.insert(0, "tests")
from w2b_testlib import write_heldout_corpus
write_heldout_corpus("/tmp/long.txt", "long_d200"); write_heldout_corpus("/tmp/v1m.txt", "heldout_v1m")
PY
RR="timeout 1500 python tests/experiments/replica_rules.py"
$RR /tmp/long.txt --sample 0.001 --size 200 --bitlevel 1 --workers 256 --positions 12288 --rules "smoothx:64:64:0:0;lib2" --out $OUT/long_d200.json 2>&1 | grep RR | tee $OUT/long_d200.txt
$RR /tmp/v1m.txt --size 512 --window 5 --negative 10 --positions 2592 --rules "smoothx:64:64:0:0;lib2" --out $OUT/v1m.json 2>&1 | grep RR | tee $OUT/v1m.txt
rm -f /tmp/long.txt /tmp/v1m.txt
echo "== done"
