#!/bin/bash
# round 2, session H: what does the two-worker workgroup cost by itself?  fused RT=13/128 VGPRs vs fused RT=25/168
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
W2B_LIB=$A $B 2>/dev/null | short "fused RT13/128: default"
W2B_LIB=$A $B --ids uniform 2>/dev/null | short "fused RT13/128: uniform"
W2B_LIB=$A $B --vocab 3700000 --dim 1000 --negative 12 2>/dev/null | short "fused RT13/128: cfg5 b1"
W2B_LIB=$A $B --dim 400 --vocab 60238 --bitlevel 2 2>/dev/null | short "fused RT13/128: cfg3 D=400 b2"
$B 2>/dev/null | short "fused RT25/168: default"
$B --loss 1 2>/dev/null | short "fused RT25/168: default, loss on"
$B --vocab 3700000 --dim 1000 --negative 12 2>/dev/null | short "fused RT25/168: cfg5 b1"
echo "== worker tests (loss slots)"
timeout 900 python -m pytest tests/test_gpu_worker.py -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | grep -E "^E  |^tests/|passed|failed|Error" | cut -c1-260
echo "== done"
