#!/bin/bash
# round 6, session q: mode 2 with the cells at ONE bit only: the exchange tests, and 8 replicas at two / four bits through the library
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r06q
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_exchange.py -q -m gpu -s 2>&1 | grep -E "EXCHANGE|passed|failed|Error|assert" | cut -c1-250 | tee $OUT/pytest_exchange.txt
python - <<'PY'
import sys
sys.path.insert(0, "tests")
from w2b_testlib import write_headline_corpus, write_heldout_corpus
write_headline_corpus("/tmp/headline.txt"); write_heldout_corpus("/tmp/long.txt", "long_d200")
PY
RR="timeout 1500 python tests/experiments/replica_rules.py"
$RR /tmp/headline.txt --bitlevel 2 --positions 672 --rules "lib2;lib2" --out $OUT/proxy_b2.json 2>&1 | grep RR | tee $OUT/proxy_b2.txt
$RR /tmp/headline.txt --bitlevel 4 --positions 672 --rules "lib2" --out $OUT/proxy_b4.json 2>&1 | grep RR | tee $OUT/proxy_b4.txt
$RR /tmp/long.txt --sample 0.001 --size 400 --bitlevel 2 --workers 256 --positions 12288 --rules "lib2;lib2" --out $OUT/long_d400b2.json 2>&1 | grep RR | tee $OUT/long_d400b2.txt
$RR /tmp/long.txt --sample 0.001 --size 400 --bitlevel 2 --workers 256 --positions 32768 --rules "lib2" --out $OUT/long_d400b2_1m.json 2>&1 | grep RR | tee $OUT/long_d400b2_1m.txt
rm -f /tmp/headline.txt /tmp/long.txt
echo "== done"
