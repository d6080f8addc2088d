#!/bin/bash
# round 6, session l: what moves the interval at which 8 replicas match one -- the row length or the workers per replica?  The two
# short-row regimes (size 200 / size 400 two bits, default sub-sampling, 100 M tokens) with 128 workers per replica (1024 in all: the
# single replica is then a full device) instead of 32, and with 32 workers per replica at shorter intervals; epoch losses are also
# read against the reference's bands (long_d200 -331.80 M, long_d400b2 -346.94 M).
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r06l
mkdir -p $OUT
python - <<'PY'
import sys
sys.path.insert(0, "tests")
from w2b_testlib import write_heldout_corpus
write_heldout_corpus("/tmp/long.txt", "long_d200")
PY
RR="timeout 1500 python tests/experiments/replica_rules.py"
$RR /tmp/long.txt --sample 0.001 --size 400 --bitlevel 2 --workers 1024 --positions 3072 --rules "lib2" --out $OUT/d400b2_w1024_p3072.json 2>&1 | grep RR | tee $OUT/d400b2_w1024_p3072.txt
$RR /tmp/long.txt --sample 0.001 --size 200 --bitlevel 1 --workers 1024 --positions 3072 --rules "lib2" --out $OUT/d200_w1024_p3072.json 2>&1 | grep RR | tee $OUT/d200_w1024_p3072.txt
$RR /tmp/long.txt --sample 0.001 --size 200 --bitlevel 1 --workers 1024 --positions 8192 --rules "lib2" --out $OUT/d200_w1024_p8192.json 2>&1 | grep RR | tee $OUT/d200_w1024_p8192.txt
$RR /tmp/long.txt --sample 0.001 --size 200 --bitlevel 1 --workers 256 --positions 6144 --rules "lib2" --out $OUT/d200_w256_p6144.json 2>&1 | grep RR | tee $OUT/d200_w256_p6144.txt
$RR /tmp/long.txt --sample 0.001 --size 200 --bitlevel 1 --workers 256 --positions 3072 --rules "lib2" --out $OUT/d200_w256_p3072.json 2>&1 | grep RR | tee $OUT/d200_w256_p3072.txt
$RR /tmp/long.txt --sample 0.001 --size 400 --bitlevel 2 --workers 256 --positions 6144 --rules "lib2" --out $OUT/d400b2_w256_p6144.json 2>&1 | grep RR | tee $OUT/d400b2_w256_p6144.txt
rm -f /tmp/long.txt
echo "== done"
