#!/bin/bash
# round 5, session c: phase timers of the row-group kernel (library built with -DW2B_PHASE_TIMERS), worker 0
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r05c
mkdir -p $OUT
export W2B_LIB=$PWD/word2bits_amd/libword2bits_hip_pt.so W2B_DEBUG=1
for shape in "d200 uniform|--vocab 60238 --dim 200 --ids uniform" "d200 zipf|--vocab 60238 --dim 200" "d800 uniform|--ids uniform" "d800 zipf|"; do
  name="${shape%%|*}"; fl="${shape##*|}"
  for arm in "groups256:row_groups=1,workers=256,loss=1" "groups512:row_groups=1,workers=512,loss=1"; do
    echo "== $name $arm" | tee -a $OUT/timers.txt
    timeout 200 python tests/experiments/arm_bench.py $fl --rounds 1 --arms "$arm" 2>&1 | grep -E "phase ticks.*\[12\]=[1-9]|^round" | tee -a $OUT/timers.txt
  done
done
echo "== done"
