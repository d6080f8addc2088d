#!/bin/bash
# round 2, session G: two workers per workgroup (software barriers), 25 rows per chunk at 168 VGPRs
set +e
export TMPDIR=/tmp
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-52s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
echo "== quick correctness"
timeout 900 python -m pytest tests/test_gpu_worker.py -m gpu -q --no-header -p no:cacheprovider --tb=short -x 2>&1 | grep -E "^E  |^tests/|passed|failed|Error" | cut -c1-260
B="timeout 300 python bench.py --cpu-baseline none --also-relaxed 0 --tokens 50000000 --steps 8 --warmup 2"
for w in 256 512 768; do W2B_HOT_ROWS=0 $B --workers $w 2>/dev/null | short "fused T25/168: workers $w (hot 0)"; done
$B 2>/dev/null | short "fused: default (hot auto)"
W2B_HOT_ROWS=2 $B 2>/dev/null | short "fused: hot 2"
W2B_HOT_ROWS=8 $B 2>/dev/null | short "fused: hot 8 (max that fits)"
$B --ids uniform 2>/dev/null | short "fused: uniform"
$B --vocab 3700000 --dim 1000 --negative 12 2>/dev/null | short "fused: cfg5 shape b1"
$B --vocab 3700000 --dim 1000 --negative 12 --bitlevel 0 2>/dev/null | short "fused: cfg5 shape b0"
$B --dim 200 --vocab 60238 2>/dev/null | short "fused: cfg1 shape D=200"
$B --dim 400 --vocab 60238 --bitlevel 2 2>/dev/null | short "fused: cfg3 shape D=400 b2"
echo "== exact + bigtable tests"
timeout 1500 python -m pytest tests/test_gpu_exact.py tests/test_gpu_bigtable.py -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | grep -E "^E  |^tests/|passed|failed|Error|s call" | cut -c1-260
echo "== done"
