#!/bin/bash
# round 2, session D: one-chunk (25 rows) sentence-resident kernel
set +e
OUT=gpurun_out/r02d
mkdir -p $OUT
export TMPDIR=/tmp
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-52s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
B="python bench.py --cpu-baseline none --also-relaxed 0 --tokens 50000000 --steps 10 --warmup 3"
timeout 600 $B 2>/dev/null | tee $OUT/b_default.json | short "default: resident 16B T=25, coherent, zipf, hot auto"
W2B_HOT_ROWS=0 timeout 600 $B 2>/dev/null | short "hot rows 0"
W2B_HOT_ROWS=2 timeout 600 $B 2>/dev/null | short "hot rows 2"
timeout 600 $B --ids uniform 2>/dev/null | short "uniform ids (hot auto = 0)"
timeout 900 $B --vocab 3700000 --dim 1000 --negative 12 2>/dev/null | tee $OUT/b_cfg5_b1.json | short "cfg5 shape b1 (V=3.7M D=1000 K=12)"
timeout 600 $B --dim 200 --vocab 60238 2>/dev/null | short "cfg1 shape D=200"
echo "== worker / exact / big-table tests"
timeout 1500 python -m pytest tests/test_gpu_worker.py tests/test_gpu_exact.py tests/test_gpu_bigtable.py -m gpu -q --no-header -p no:cacheprovider --tb=short -x 2>&1 | grep -E "^E  |^tests/|passed|failed|Error|s call" | cut -c1-260
echo "== done"
