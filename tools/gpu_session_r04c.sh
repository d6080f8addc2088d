#!/bin/bash
# round 4, session c: the candidate defaults -- context rows updated by lossless atomic adds (automatic prefix), no per-XCD
# copies of context rows, plain kernel everywhere -- against the number of per-XCD copies of TARGET rows and their merge
# period: tests of what changed, throughput (with / without the table prefetch), fidelity in every regime, the exchange.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r04c
mkdir -p $OUT
echo "== (1) tests"
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_worker.py tests/test_gpu_parity.py tests/test_gpu_exact.py tests/test_gpu_cli.py tests/test_gpu_exchange.py -k "not training_effect" 2>&1 | tail -15 | tee $OUT/pytest_quick.txt
echo "== (2) arms: throughput"
A="default:;noloss:loss=0;v0:hot_rows_v=0;v4:hot_rows_v=4;v8:hot_rows_v=8;v16:hot_rows_v=16;v32:hot_rows_v=32;v4p1:hot_rows_v=4,hot_period=1;v8p1:hot_rows_v=8,hot_period=1;v16p1:hot_rows_v=16,hot_period=1;vauto_p1:hot_period=1;r03like:hot_rows_u=113,atomic_rank_u=-1;resident:window_cache=1"
timeout 900 python tests/experiments/arm_bench.py --rounds 2 --out $OUT/arm_bench.json --arms "$A" 2>&1 | grep -v amdgpu.ids | tee $OUT/arm_bench.txt
echo "== (2b) without the table prefetch"
W2B_LIB=$PWD/word2bits_amd/libword2bits_hip_nopf.so timeout 600 python tests/experiments/arm_bench.py --rounds 2 --arms "nopf default:;nopf noloss:loss=0;nopf v8:hot_rows_v=8" 2>&1 | grep -v amdgpu.ids | tee $OUT/arm_bench_nopf.txt
echo "== (3) fidelity matrix, benchmarked regime"
ARMS="default:;v0:-hot-rows-v 0;v4:-hot-rows-v 4;v8:-hot-rows-v 8;v16:-hot-rows-v 16;v4p1:-hot-rows-v 4 -hot-period 1;v8p1:-hot-rows-v 8 -hot-period 1;v16p1:-hot-rows-v 16 -hot-period 1;vauto_p1:-hot-period 1;v8w1000:-hot-rows-v 8 -hot-weight 1000"
timeout 1500 python tests/experiments/fidelity_matrix.py --jobs headline --threads 64,256,0 --kernel auto --out $OUT/fidelity.jsonl --arms "$ARMS" 2>&1 | tee $OUT/fidelity_headline.txt | cut -c1-220
echo "== (4) the other regimes"
timeout 900 python tests/experiments/fidelity_matrix.py --jobs heldout_zipf12,heldout_k5,text8size --threads 0,256,64 --kernel auto --out $OUT/fidelity.jsonl --arms "default:;v0:-hot-rows-v 0;v8p1:-hot-rows-v 8 -hot-period 1" 2>&1 | tee $OUT/fidelity_other.txt | cut -c1-220
echo "== (5) exchange matrix"
timeout 1200 python tests/experiments/exchange_matrix.py --positions 1024,256,64 --replicas 2,4 --tiers 8:4,8:16 > $OUT/exchange_matrix.txt 2>$OUT/exchange_matrix.err
grep XM $OUT/exchange_matrix.txt; tail -5 $OUT/exchange_matrix.err
echo "== done"
