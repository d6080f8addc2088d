#!/bin/bash
# round 6, session p: at two bits the quantization-cell criterion ends -9.6 % on the configs[1]-shape proxy (session o).  Is the SIGN
# alone (the harness's signsafe rule) the better criterion there -- and does it keep the two-bit long stream (-1.4 % with cells)?
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r06p
mkdir -p $OUT
python - <<'PY'
import sys
sys.path.insert(0, "tests")
from w2b_testlib import write_headline_corpus, write_heldout_corpus
write_headline_corpus("/tmp/headline.txt"); write_heldout_corpus("/tmp/long.txt", "long_d200")
PY
RR="timeout 1500 python tests/experiments/replica_rules.py"
$RR /tmp/headline.txt --bitlevel 2 --positions 672 --rules "signsafe:64:64:0:0:1.0;smoothx:64:64:0:0;lib2" --out $OUT/proxy_b2.json 2>&1 | grep RR | tee $OUT/proxy_b2.txt
$RR /tmp/headline.txt --bitlevel 4 --positions 672 --rules "signsafe:64:64:0:0:1.0;lib2" --out $OUT/proxy_b4.json 2>&1 | grep RR | tee $OUT/proxy_b4.txt
$RR /tmp/long.txt --sample 0.001 --size 400 --bitlevel 2 --workers 256 --positions 12288 --rules "signsafe:64:64:0:0:1.0;lib2" --out $OUT/long_d400b2.json 2>&1 | grep RR | tee $OUT/long_d400b2.txt
rm -f /tmp/headline.txt /tmp/long.txt
echo "== done"
