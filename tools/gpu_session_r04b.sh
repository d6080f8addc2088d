#!/bin/bash
# round 4, session b: (1) how many atomic adds ONE row takes; (2) throughput of the knob arms, this build and the round-3
# library side by side (W2B_LIB); (3) fidelity of the benchmarked regime without per-XCD copies / with lossless rows /
# other consensus weights; (4) the held-out steep-Zipf regime, where round 3's defaults are 13 % off with the
# sentence-resident kernel; (5) the exchange: launch length x scheme on one GPU.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r04b
mkdir -p $OUT
echo "== (1) atomic probe"
timeout 180 tools/atomic_probe 2>&1 | grep -E "rows" | tee $OUT/atomic_probe.txt
echo "== (2) arms: throughput (this build)"
timeout 900 python tests/experiments/arm_bench.py --rounds 2 --out $OUT/arm_bench.json --arms "default:;noloss:loss=0;late:hot_late=1;fresh128:fresh_rank_u=128;fresh2000:fresh_rank_u=2000;atomu300:atomic_rank_u=300,atomic_rank=0;atomu3000:atomic_rank_u=3000,atomic_rank=0;atomv300:atomic_rank=300,atomic_rank_u=-1;atomv1000:atomic_rank=1000,atomic_rank_u=-1;atomuv:atomic_rank=1000,atomic_rank_u=3000;hot16+atomuv:hot_rows_u=16,hot_rows_v=16,atomic_rank=1000,atomic_rank_u=3000;hot0+atomuv:hot_rows_u=0,hot_rows_v=0,atomic_rank=1000,atomic_rank_u=3000;p8:hot_period=8;resident:window_cache=1" 2>&1 | grep -v amdgpu.ids | tee $OUT/arm_bench.txt
if [ -f word2bits_amd/libword2bits_hip_r03.so ]; then
echo "== (2b) the round-3 library (W2B_LIB): default / loss"
W2B_LIB_ALLOW_MISSING=1 W2B_LIB=$PWD/word2bits_amd/libword2bits_hip_r03.so timeout 600 python tests/experiments/arm_bench.py --rounds 2 --arms "r03 noloss:loss=0;r03 loss:loss=1" 2>&1 | grep -v amdgpu.ids | tee $OUT/arm_bench_r03.txt
fi
echo "== (3) fidelity matrix, benchmarked regime"
ARMS="default:;hot0:-hot-rows 0;hot0+atomuv:-hot-rows 0 -atomic-rank 1000 -atomic-rank-u 3000;hot0+atomu:-hot-rows 0 -atomic-rank 0 -atomic-rank-u 3000;hot0+atomv:-hot-rows 0 -atomic-rank 1000 -atomic-rank-u -1;hotu0:-hot-rows-u 0;hotv0:-hot-rows-v 0;hotu0+atomu:-hot-rows-u 0 -atomic-rank 0 -atomic-rank-u 3000;hotv0+atomv:-hot-rows-v 0 -atomic-rank 1000 -atomic-rank-u -1;w500:-hot-weight 500;w1000:-hot-weight 1000;p1:-hot-period 1;hot16+atomuv:-hot-rows 16 -atomic-rank 1000 -atomic-rank-u 3000"
timeout 1500 python tests/experiments/fidelity_matrix.py --jobs headline --threads 64,256,0 --kernel plain --out $OUT/fidelity.jsonl --arms "$ARMS" 2>&1 | tee $OUT/fidelity_headline.txt | cut -c1-220
echo "== (4) held-out regimes and the text8-sized one: consensus rows of the sentence-resident kernel"
ARMS4="default:;hotu0:-hot-rows-u 0;hot0:-hot-rows 0;refresh0:-window-refresh 0;w1000:-hot-weight 1000;hot0+atomuv:-hot-rows 0 -atomic-rank 1000 -atomic-rank-u 3000"
timeout 900 python tests/experiments/fidelity_matrix.py --jobs heldout_zipf12,heldout_k5,text8size --threads 0,256 --kernel both --out $OUT/fidelity.jsonl --arms "$ARMS4" 2>&1 | tee $OUT/fidelity_other.txt | cut -c1-220
echo "== (5) exchange matrix"
timeout 1200 python tests/experiments/exchange_matrix.py --positions 1024,256,64 --replicas 2,4 --tiers 8:4,8:16 2>&1 | grep XM | tee $OUT/exchange_matrix.txt
echo "== done"
