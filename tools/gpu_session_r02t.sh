#!/bin/bash
set +e
export TMPDIR=/tmp
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-40s %8.2f Mw/s  frac %.3f' % (sys.argv[1], d['value']/1e6, d['roofline']['frac']))
" "$1"; }
B="timeout 300 python bench.py --cpu-baseline none --also-relaxed 0 --also-legs 0 --tokens 50000000 --steps 8 --warmup 2"
$B 2>/dev/null | short "default"
W2B_HOT_ROWS=0 $B 2>/dev/null | short "hot 0"
$B --ids uniform 2>/dev/null | short "uniform"
$B --vocab 3700000 --dim 1000 --negative 12 2>/dev/null | short "cfg5 b1"
timeout 900 python -m pytest tests/test_gpu_worker.py tests/test_gpu_exact.py -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | grep -E "^E  |^tests/|passed|failed|Error" | cut -c1-260
