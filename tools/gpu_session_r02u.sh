#!/bin/bash
set +e
export TMPDIR=/tmp
L=$PWD/word2bits_amd/libword2bits_hip_t.so
B="timeout 300 python bench.py --cpu-baseline none --also-relaxed 0 --also-legs 0 --tokens 50000000 --steps 8 --warmup 2"
for d in 200 400; do
echo "-- D=$d V=60238"; W2B_DEBUG=1 W2B_LIB=$L $B --dim $d --vocab 60238 2>&1 | grep -E "phase ticks|resident kernel|^\{" | cut -c1-420 | tail -3
done
echo "-- D=200 V=400000"; W2B_DEBUG=1 W2B_LIB=$L $B --dim 200 2>&1 | grep -E "phase ticks|resident kernel|^\{" | cut -c1-420 | tail -3
