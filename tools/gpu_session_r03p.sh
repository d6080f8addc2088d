#!/bin/bash
# round 3, session p: automatic atomic row updates only for cache-sized tables -- uniform ids at size 200, planted gates
export TMPDIR=/tmp
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); wk=d['config'].get('worker_kernel') or {}
        print('%-40s %8.2f Mw/s  frac %.3f  launch %.2f ms  %s hot %s workers %s' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'], wk.get('hot_rows_with_xcd_copies'), d['config'].get('workers')))
" "$1"; }
B="python bench.py --cpu-baseline none --also-relaxed 0 --also-legs 0 --also-shapes 0 --tokens 30000000 --steps 10 --warmup 3"
run() { name="$1"; shift; timeout 300 $B "$@" 2>/dev/null | short "$name"; }
run "d200 uniform" --dim 200 --vocab 60238 --ids uniform
run "d400 b2 uniform" --dim 400 --vocab 60238 --bitlevel 2 --ids uniform
run "cfg2 uniform" --ids uniform
timeout 600 python -m pytest tests/test_gpu_fidelity.py -k "planted or warned" -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -4
