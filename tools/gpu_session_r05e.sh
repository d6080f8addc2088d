#!/bin/bash
# round 5, session e: refreshed read copies of the hottest context rows (row-group kernel): parity suite, then throughput
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r05e
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_groups.py -x -q -m gpu > $OUT/pytest_groups.txt 2>&1; tail -3 $OUT/pytest_groups.txt
A="g256_auto:row_groups=1,workers=256,loss=1;g256_none:row_groups=1,workers=256,loss=1,refresh_rows_u=-1;g256_rc4:row_groups=1,workers=256,loss=1,refresh_rows_u=4;g256_rc64:row_groups=1,workers=256,loss=1,refresh_rows_u=64;g512_auto:row_groups=1,workers=512,loss=1;g64_auto:row_groups=1,workers=64,loss=1;g64_none:row_groups=1,workers=64,loss=1,refresh_rows_u=-1"
echo "== d200" | tee -a $OUT/rc.txt
timeout 300 python tests/experiments/arm_bench.py --vocab 60238 --dim 200 --rounds 1 --arms "$A" 2>&1 | grep -A9 "== best" | tee -a $OUT/rc.txt
echo "== d400 b2" | tee -a $OUT/rc.txt
timeout 300 python tests/experiments/arm_bench.py --vocab 60238 --dim 400 --bitlevel 2 --rounds 1 --arms "$A" 2>&1 | grep -A9 "== best" | tee -a $OUT/rc.txt
echo "== d800" | tee -a $OUT/rc.txt
timeout 300 python tests/experiments/arm_bench.py --rounds 1 --arms "g256_auto:row_groups=1,workers=256,loss=1;g256_none:row_groups=1,workers=256,loss=1,refresh_rows_u=-1" 2>&1 | grep -A4 "== best" | tee -a $OUT/rc.txt
echo "== done"
