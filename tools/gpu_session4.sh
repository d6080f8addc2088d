#!/bin/bash
set +e
OUT=gpurun_out/s4
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | grep -E "^E  |^tests/|passed|failed|Error" | cut -c1-260 | tee $OUT/pytest.log
B="python bench.py --cpu-baseline none"
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-34s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
timeout 600 $B 2>/dev/null | short "tuples zipf coherent" | tee -a $OUT/variants.log
timeout 600 $B --relaxed 1 2>/dev/null | short "tuples zipf relaxed" | tee -a $OUT/variants.log
timeout 600 $B --ids uniform 2>/dev/null | short "tuples uniform coherent" | tee -a $OUT/variants.log
timeout 600 $B --form worker 2>/dev/null | short "worker zipf coherent" | tee -a $OUT/variants.log
timeout 600 $B --form worker --relaxed 1 2>/dev/null | short "worker zipf relaxed" | tee -a $OUT/variants.log
timeout 600 $B --form worker --ids uniform 2>/dev/null | short "worker uniform coherent" | tee -a $OUT/variants.log
echo "== accuracy b1 (coherent default)"; timeout 1200 python tools/accuracy_experiment.py --bitlevel 1 --cpu-threads 8,64 --gpu-threads 8,64,1024 --variants coherent 2>&1 | cut -c1-330 | tee $OUT/acc_b1.log
echo "== done"
