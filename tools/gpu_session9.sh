#!/bin/bash
set +e
export TMPDIR=/tmp
B="python bench.py --cpu-baseline none --also-relaxed 0 --tokens 30000000 --steps 6 --warmup 2 --form worker --positions 1024"
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-40s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
for w in 128 256 384 512 768; do timeout 300 $B --workers $w 2>/dev/null | short "wc=1 workers=$w"; done
