#!/bin/bash
set +e
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_fidelity.py tests/test_gpu_integration.py tests/test_gpu_worker.py -m gpu -q --no-header -p no:cacheprovider --tb=short -s 2>&1 | grep -E "^E  |^tests/|passed|failed|Error|FIDELITY" | cut -c1-330
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-40s %8.2f Mw/s  frac %.3f' % (sys.argv[1], d['value']/1e6, d['roofline']['frac']), d['config']['worker_kernel'])
" "$1"; }
B="timeout 300 python bench.py --cpu-baseline none --also-relaxed 0 --also-legs 0 --tokens 50000000 --steps 8 --warmup 2"
$B 2>/dev/null | short "bench default (hot <= 4, period 8)"
W2B_HOT_PERIOD=32 W2B_HOT_ROWS=6 $B 2>/dev/null | short "bench hot 6 period 32"
W2B_HOT_ROWS=0 $B 2>/dev/null | short "bench hot 0"
