#!/bin/bash
# round 5, session j: late target loads + LDS read-ahead: parity, throughput (uniform / Zipf, copies none / 4 / auto), timers
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r05j
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_groups.py -x -q -m gpu > $OUT/pytest_groups.txt 2>&1; tail -3 $OUT/pytest_groups.txt
A="g256_none:row_groups=1,workers=256,loss=1,refresh_rows_u=-1;g256_rc4:row_groups=1,workers=256,loss=1,refresh_rows_u=4;g256_auto:row_groups=1,workers=256,loss=1;g128_rc4:row_groups=1,workers=128,loss=1,refresh_rows_u=4;g512_rc4:row_groups=1,workers=512,loss=1,refresh_rows_u=4"
for shape in "d200 zipf|--vocab 60238 --dim 200" "d200 uniform|--vocab 60238 --dim 200 --ids uniform" "d400b2 zipf|--vocab 60238 --dim 400 --bitlevel 2" "d800 zipf|"; do
  name="${shape%%|*}"; fl="${shape##*|}"
  echo "== $name" | tee -a $OUT/bench.txt
  timeout 300 python tests/experiments/arm_bench.py $fl --rounds 1 --arms "$A" 2>&1 | grep -A6 "== best" | tee -a $OUT/bench.txt
done
export W2B_LIB=$PWD/word2bits_amd/libword2bits_hip_pt.so W2B_DEBUG=1
for shape in "d200 uniform|--vocab 60238 --dim 200 --ids uniform" "d200 zipf|--vocab 60238 --dim 200"; do
  name="${shape%%|*}"; fl="${shape##*|}"
  for arm in "g256_none:row_groups=1,workers=256,loss=1,refresh_rows_u=-1" "g256_rc4:row_groups=1,workers=256,loss=1,refresh_rows_u=4"; do
    echo "== $name $arm" | tee -a $OUT/timers.txt
    timeout 200 python tests/experiments/arm_bench.py $fl --rounds 1 --arms "$arm" 2>&1 | grep -E "phase ticks.*\[12\]=[1-9]|^round" | tee -a $OUT/timers.txt
  done
done
echo "== done"
