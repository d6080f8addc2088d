#!/bin/bash
# round 2, session A: memory-shape probe + large-table (per-row descriptor) form under the parity tests + baseline matrix
set +e
OUT=gpurun_out/r02a
mkdir -p $OUT
export TMPDIR=/tmp
echo "== row probe"; timeout 300 tools/row_probe 2>&1 | tee $OUT/row_probe.txt
echo "== parity tests with W2B_FORCE_ROW_DESC=1 (large-table form on small tables)"
W2B_FORCE_ROW_DESC=1 timeout 900 python -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py tests/test_gpu_worker.py -m gpu -q --no-header -p no:cacheprovider --tb=short -x 2>&1 | grep -E "^E  |^tests/|passed|failed|Error" | cut -c1-260
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-52s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
B="python bench.py --cpu-baseline none --also-relaxed 0 --tokens 50000000 --steps 10 --warmup 3"
timeout 600 $B 2>/dev/null | tee $OUT/b_default.json | short "default (coherent, resident, zipf)"
W2B_HOT_ROWS=0 timeout 600 $B 2>/dev/null | short "default, hot rows off"
timeout 600 $B --ids uniform 2>/dev/null | short "uniform ids"
W2B_HOT_ROWS=0 timeout 600 $B --ids uniform 2>/dev/null | short "uniform ids, hot rows off"
W2B_FORCE_ROW_DESC=1 timeout 600 $B 2>/dev/null | short "default, per-row descriptors"
timeout 600 $B --relaxed 1 --window-cache 1 2>/dev/null | short "resident relaxed"
timeout 900 $B --vocab 3700000 --dim 1000 --negative 12 2>/dev/null | tee $OUT/b_cfg5_b1.json | short "cfg5 shape b1 (V=3.7M D=1000 K=12)"
timeout 900 $B --vocab 3700000 --dim 1000 --negative 12 --bitlevel 0 2>/dev/null | tee $OUT/b_cfg5_b0.json | short "cfg5 shape b0"
echo "== done"
