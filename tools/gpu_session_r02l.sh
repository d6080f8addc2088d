#!/bin/bash
set +e
export TMPDIR=/tmp
L=$PWD/word2bits_amd/libword2bits_hip_t.so
B="timeout 300 python bench.py --cpu-baseline none --also-relaxed 0 --also-legs 0 --tokens 50000000 --steps 8 --warmup 2"
echo "-- cfg2 512 workers"; W2B_DEBUG=1 W2B_LIB=$L $B 2>&1 | grep -E "phase ticks|resident kernel" | tail -3
echo "-- cfg2 256 workers"; W2B_DEBUG=1 W2B_LIB=$L $B --workers 256 2>&1 | grep -E "phase ticks" | tail -1
echo "-- cfg5 512 workers"; W2B_DEBUG=1 W2B_LIB=$L $B --vocab 3700000 --dim 1000 --negative 12 2>&1 | grep -E "phase ticks" | tail -1
echo "-- cfg2 uniform"; W2B_DEBUG=1 W2B_LIB=$L $B --ids uniform 2>&1 | grep -E "phase ticks" | tail -1
