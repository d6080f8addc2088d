#!/bin/bash
# round 2, session E: is a second 5-wavefront workgroup really resident per CU at 168 VGPRs?
set +e
export TMPDIR=/tmp
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-52s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
B="python bench.py --cpu-baseline none --also-relaxed 0 --tokens 50000000 --steps 8 --warmup 2"
for w in 128 256 384 512 768; do W2B_HOT_ROWS=0 timeout 600 $B --workers $w 2>/dev/null | short "workers $w (hot 0)"; done
W2B_HOT_ROWS=0 timeout 600 $B --workers 256 --ids uniform 2>/dev/null | short "workers 256 uniform"
W2B_HOT_ROWS=0 timeout 600 $B --workers 512 --ids uniform 2>/dev/null | short "workers 512 uniform"
echo "== done"
