#!/bin/bash
set +e
export TMPDIR=/tmp
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-52s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
B="timeout 300 python bench.py --cpu-baseline none --also-relaxed 0 --also-legs 0 --tokens 50000000 --steps 8 --warmup 2"
for v in c m c m; do
  L=$PWD/word2bits_amd/libword2bits_hip_$v.so
  W2B_LIB=$L $B 2>/dev/null | short "variant $v: default"
  W2B_LIB=$L $B --ids uniform 2>/dev/null | short "variant $v: uniform"
done
L=$PWD/word2bits_amd/libword2bits_hip_m.so
W2B_LIB=$L W2B_HOT_ROWS=0 $B 2>/dev/null | short "m: zipf, hot 0"
W2B_LIB=$L W2B_HOT_ROWS=0 $B --zipf-shift 6 2>/dev/null | short "m: zipf without its 6 hottest words, hot 0"
W2B_LIB=$L W2B_HOT_ROWS=0 $B --zipf-shift 100 2>/dev/null | short "m: zipf without its 100 hottest words, hot 0"
W2B_LIB=$L W2B_HOT_ROWS=0 $B --zipf-shift 1000 2>/dev/null | short "m: zipf without its 1000 hottest words, hot 0"
W2B_LIB=$L $B --vocab 3700000 --dim 1000 --negative 12 2>/dev/null | short "m: cfg5 b1"
echo "== done"
