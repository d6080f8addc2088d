#!/bin/bash
# round 4, session d: per-XCD copies only where a row's update rate exceeds what one row takes (few rows, full GPU only),
# lossless context rows below them: throughput and fidelity at 1024 / 256 workers; the exchange matrix.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r04d
mkdir -p $OUT
A="r03like:hot_rows_u=113,hot_rows_v=113,atomic_rank_u=-1;c12_8:hot_rows_u=12,hot_rows_v=8;c12_8p2:hot_rows_u=12,hot_rows_v=8,hot_period=2;c12_8p1:hot_rows_u=12,hot_rows_v=8,hot_period=1;c12_8na:hot_rows_u=12,hot_rows_v=8,atomic_rank_u=-1;c32_16:hot_rows_u=32,hot_rows_v=16;c32_16p2:hot_rows_u=32,hot_rows_v=16,hot_period=2;c64_32:hot_rows_u=64,hot_rows_v=32;c113_113:hot_rows_u=113,hot_rows_v=113;c12_0:hot_rows_u=12,hot_rows_v=0;c6_4:hot_rows_u=6,hot_rows_v=4;faithful:hot_rows_u=0,hot_rows_v=0;auto:"
echo "== (1) throughput at the headline shape, full GPU"
timeout 900 python tests/experiments/arm_bench.py --tokens 60000000 --rounds 2 --out $OUT/arm_bench.json --arms "$A" 2>&1 | grep -v amdgpu.ids | tee $OUT/arm_bench.txt
echo "== (2) fidelity, benchmarked regime, 1024 workers"
F="r03like:-hot-rows-u 113 -hot-rows-v 113 -atomic-rank-u -1;c12_8:-hot-rows-u 12 -hot-rows-v 8;c12_8p2:-hot-rows-u 12 -hot-rows-v 8 -hot-period 2;c12_8p1:-hot-rows-u 12 -hot-rows-v 8 -hot-period 1;c12_8na:-hot-rows-u 12 -hot-rows-v 8 -atomic-rank-u -1;c32_16:-hot-rows-u 32 -hot-rows-v 16;c32_16p2:-hot-rows-u 32 -hot-rows-v 16 -hot-period 2;c64_32:-hot-rows-u 64 -hot-rows-v 32;c113_113:-hot-rows-u 113 -hot-rows-v 113;c12_0:-hot-rows-u 12 -hot-rows-v 0;c6_4:-hot-rows-u 6 -hot-rows-v 4;faithful:-hot-rows-u 0 -hot-rows-v 0"
timeout 900 python tests/experiments/fidelity_matrix.py --jobs headline --threads 1024 --kernel auto --out $OUT/fidelity.jsonl --arms "$F" 2>&1 | tee $OUT/fidelity_1024.txt | cut -c1-200
echo "== (3) fidelity, 256 workers: what the rate rule gives there (3 / 1 copies) and neighbours"
G="c3_1:-hot-rows-u 3 -hot-rows-v 1;c3_1p1:-hot-rows-u 3 -hot-rows-v 1 -hot-period 1;c3_0:-hot-rows-u 3 -hot-rows-v 0;c6_2:-hot-rows-u 6 -hot-rows-v 2;c0_0:-hot-rows-u 0 -hot-rows-v 0"
timeout 600 python tests/experiments/fidelity_matrix.py --jobs headline --threads 256 --kernel auto --out $OUT/fidelity.jsonl --arms "$G" 2>&1 | tee $OUT/fidelity_256.txt | cut -c1-200
echo "== (4) exchange matrix"
timeout 1200 python tests/experiments/exchange_matrix.py --positions 1024,256,64 --replicas 2,4 --tiers 8:4,8:16 > $OUT/exchange_matrix.txt 2>$OUT/exchange_matrix.err
grep XM $OUT/exchange_matrix.txt; tail -3 $OUT/exchange_matrix.err
echo "== done"
