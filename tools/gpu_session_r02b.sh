#!/bin/bash
# round 2, session B: 16-byte-column sentence-resident kernel -- parity tests, then the bench matrix against the 8-byte form
set +e
OUT=gpurun_out/r02b
mkdir -p $OUT
export TMPDIR=/tmp
echo "== worker/exact tests (16-byte columns are now the default of the resident kernel)"
timeout 900 python -m pytest tests/test_gpu_worker.py tests/test_gpu_exact.py -m gpu -q --no-header -p no:cacheprovider --tb=short -x 2>&1 | grep -E "^E  |^tests/|passed|failed|Error" | cut -c1-260
echo "== torch view test alone (failed in session A under W2B_FORCE_ROW_DESC)"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider --tb=long -x -k "model_tensor" 2>&1 | tail -30 | cut -c1-200
echo "== big-table tests"
timeout 1500 python -m pytest tests/test_gpu_bigtable.py -m gpu -q --no-header -p no:cacheprovider --tb=short --durations=8 2>&1 | grep -E "^E  |^tests/|passed|failed|Error|s call" | cut -c1-260
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-52s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
B="python bench.py --cpu-baseline none --also-relaxed 0 --tokens 50000000 --steps 10 --warmup 3"
timeout 600 $B 2>/dev/null | tee $OUT/b_default.json | short "default: 16B columns, coherent, zipf"
W2B_WIN2_VEC=2 timeout 600 $B 2>/dev/null | short "8B columns (round-1 shape)"
W2B_HOT_ROWS=0 timeout 600 $B 2>/dev/null | short "16B, hot rows off"
timeout 600 $B --ids uniform 2>/dev/null | short "16B, uniform ids"
W2B_HOT_ROWS=0 timeout 600 $B --ids uniform 2>/dev/null | short "16B, uniform ids, hot rows off"
timeout 600 $B --relaxed 1 --window-cache 1 2>/dev/null | short "16B, resident relaxed"
W2B_FORCE_ROW_DESC=1 timeout 600 $B 2>/dev/null | short "16B, per-row descriptors"
timeout 600 $B --dim 200 --vocab 60238 2>/dev/null | short "cfg1 shape D=200"
timeout 600 $B --dim 400 --vocab 60238 --bitlevel 2 2>/dev/null | short "cfg3 shape D=400 b2"
echo "== done"
