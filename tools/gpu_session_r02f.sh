#!/bin/bash
# round 2, session F: 13-row chunks at 128 VGPRs (two workers per CU) vs 25-row chunks at 168 (one)
set +e
export TMPDIR=/tmp
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-52s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
B="python bench.py --cpu-baseline none --also-relaxed 0 --tokens 50000000 --steps 8 --warmup 2"
for w in 256 512 768; do W2B_HOT_ROWS=0 timeout 600 $B --workers $w 2>/dev/null | short "T13/128: workers $w (hot 0)"; done
timeout 600 $B 2>/dev/null | short "T13/128: default (hot auto)"
W2B_HOT_ROWS=2 timeout 600 $B 2>/dev/null | short "T13/128: hot 2"
timeout 600 $B --ids uniform 2>/dev/null | short "T13/128: uniform"
timeout 900 $B --vocab 3700000 --dim 1000 --negative 12 2>/dev/null | short "T13/128: cfg5 shape b1"
timeout 600 $B --dim 200 --vocab 60238 2>/dev/null | short "T13/128: cfg1 shape D=200"
timeout 600 $B --dim 400 --vocab 60238 --bitlevel 2 2>/dev/null | short "T13/128: cfg3 shape D=400 b2"
echo "== tests"
timeout 1500 python -m pytest tests/test_gpu_worker.py tests/test_gpu_exact.py tests/test_gpu_bigtable.py -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | grep -E "^E  |^tests/|passed|failed|Error|s call" | cut -c1-260
echo "== done"
