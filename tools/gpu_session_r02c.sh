#!/bin/bash
# round 2, session C: rewritten sentence-resident kernel (16-byte columns, scratch entries, LDS hot rows)
set +e
OUT=gpurun_out/r02c
mkdir -p $OUT
export TMPDIR=/tmp
echo "== worker / exact / big-table tests"
timeout 1500 python -m pytest tests/test_gpu_worker.py tests/test_gpu_exact.py tests/test_gpu_bigtable.py tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider --tb=short --durations=5 2>&1 | grep -E "^E  |^tests/|passed|failed|Error|s call" | cut -c1-260
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-52s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
B="python bench.py --cpu-baseline none --also-relaxed 0 --tokens 50000000 --steps 10 --warmup 3"
timeout 600 $B 2>/dev/null | tee $OUT/b_default.json | short "default: resident 16B, coherent, zipf, hot auto"
W2B_HOT_ROWS=0 timeout 600 $B 2>/dev/null | short "hot rows 0"
W2B_HOT_ROWS=2 timeout 600 $B 2>/dev/null | short "hot rows 2"
W2B_HOT_ROWS=8 timeout 600 $B 2>/dev/null | short "hot rows 8 (as many as fit)"
timeout 600 $B --ids uniform 2>/dev/null | short "uniform ids (hot auto = 0)"
W2B_FORCE_ROW_DESC=1 timeout 600 $B 2>/dev/null | short "per-row descriptors"
timeout 600 $B --window-cache 0 2>/dev/null | short "plain kernel coherent"
timeout 600 $B --dim 200 --vocab 60238 2>/dev/null | short "cfg1 shape D=200"
timeout 600 $B --dim 400 --vocab 60238 --bitlevel 2 2>/dev/null | short "cfg3 shape D=400 b2"
timeout 900 $B --vocab 3700000 --dim 1000 --negative 12 2>/dev/null | tee $OUT/b_cfg5_b1.json | short "cfg5 shape b1 (V=3.7M D=1000 K=12)"
timeout 900 $B --vocab 3700000 --dim 1000 --negative 12 --bitlevel 0 2>/dev/null | tee $OUT/b_cfg5_b0.json | short "cfg5 shape b0"
echo "== done"
