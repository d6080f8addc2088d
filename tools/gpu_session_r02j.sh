#!/bin/bash
# round 2, session J: LDS-counter barrier (a) vs the same + vmcnt(0) drain (b) vs s_barrier (c); all WPG=1, 13 rows, 128 VGPRs
set +e
export TMPDIR=/tmp
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-52s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
B="timeout 300 python bench.py --cpu-baseline none --also-relaxed 0 --also-legs 0 --tokens 50000000 --steps 8 --warmup 2"
for v in a b c a b c; do
  L=$PWD/word2bits_amd/libword2bits_hip_$v.so
  W2B_LIB=$L $B 2>/dev/null | short "variant $v: default"
  W2B_LIB=$L $B --vocab 3700000 --dim 1000 --negative 12 2>/dev/null | short "variant $v: cfg5 b1"
done
echo "== done"
