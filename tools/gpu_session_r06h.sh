#!/bin/bash
# round 6, session h: (1) the planted corpus (small flat vocabulary) with fewer workers AT ONCE (w2b_tuning.concurrent_workers): can
# the one stated exception of the 1.5 % floor (configs[2] shape, 64 workers) be retired?  (2) 8 replicas at the row-group kernel's
# row lengths on the 100 M-token long stream WITH the reference's default sub-sampling, so that the epoch losses can be read against
# the reference's own bands (long_d200: -331.80 M, long_d400b2: -346.94 M).
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r06h
mkdir -p $OUT
echo "== (1) planted corpus: workers at once"
timeout 900 python tests/experiments/planted_concurrency.py --out $OUT/planted_concurrency.json 2>&1 | grep PC | tee $OUT/planted_concurrency.txt
python - <<'PY'
import sys
sys.path.insert(0, "tests")
from w2b_testlib import write_heldout_corpus
write_heldout_corpus("/tmp/long.txt", "long_d200")
PY
echo "== (2) 8 replicas x 32 workers, default sub-sampling, against the reference bands"
RR="timeout 1500 python tests/experiments/replica_rules.py"
$RR /tmp/long.txt --sample 0.001 --size 400 --bitlevel 2 --workers 256 --positions 12288 --rules "lib2;lib2" --out $OUT/rules_long_d400b2.json 2>&1 | grep RR | tee $OUT/rules_long_d400b2.txt
$RR /tmp/long.txt --sample 0.001 --size 200 --bitlevel 1 --workers 256 --positions 12288 --rules "lib2;lib2" --out $OUT/rules_long_d200.json 2>&1 | grep RR | tee $OUT/rules_long_d200.txt
$RR /tmp/long.txt --sample 0.001 --size 400 --bitlevel 2 --workers 256 --positions 32768 --rules "lib2" --out $OUT/rules_long_d400b2_1m.json 2>&1 | grep RR | tee $OUT/rules_long_d400b2_1m.txt
rm -f /tmp/long.txt
echo "== done"
