#!/bin/bash
# round 2, session I: one worker per workgroup, 13 rows / 128 VGPRs, per-worker LDS barriers (vs s_barrier in session F: 0.691 / 0.859)
set +e
export TMPDIR=/tmp
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-52s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
B="timeout 300 python bench.py --cpu-baseline none --also-relaxed 0 --also-legs 0 --tokens 50000000 --steps 8 --warmup 2"
A=$PWD/word2bits_amd/libword2bits_hip_a.so
W2B_LIB=$A $B 2>/dev/null | short "WPG1 RT13/128 sw-barrier: default"
W2B_LIB=$A $B --loss 1 2>/dev/null | short "WPG1 RT13/128 sw-barrier: default, loss on"
W2B_LIB=$A $B --bitlevel 2 2>/dev/null | short "WPG1 RT13/128 sw-barrier: bitlevel 2"
W2B_LIB=$A $B --ids uniform 2>/dev/null | short "WPG1 RT13/128 sw-barrier: uniform"
W2B_LIB=$A $B --vocab 3700000 --dim 1000 --negative 12 2>/dev/null | short "WPG1 RT13/128 sw-barrier: cfg5 b1"
W2B_LIB=$A $B --vocab 3700000 --dim 1000 --negative 12 --bitlevel 0 2>/dev/null | short "WPG1 RT13/128 sw-barrier: cfg5 b0"
W2B_LIB=$A $B --dim 400 --vocab 60238 --bitlevel 2 2>/dev/null | short "WPG1 RT13/128 sw-barrier: cfg3 D=400 b2"
W2B_LIB=$A $B --dim 200 --vocab 60238 2>/dev/null | short "WPG1 RT13/128 sw-barrier: cfg1 D=200"
echo "== done"
