#!/bin/bash
# round 4, session e: the final policy build (copies only on a full device, lossless context rows below it, plain kernel):
# fidelity of every regime incl. the planted ones and the full-device case; exchange: saturation threshold x scheme.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r04e
mkdir -p $OUT
echo "== (1) fidelity, final policy"
timeout 900 python tests/experiments/fidelity_matrix.py --jobs planted_b1_d200,planted_cfg2_b2_d400 --threads 8,64 --kernel both --out $OUT/fidelity.jsonl --arms "default:" 2>&1 | tee $OUT/fidelity_planted.txt | cut -c1-220
timeout 900 python tests/experiments/fidelity_matrix.py --jobs headline --threads 0,1024,768,256,64 --kernel auto --out $OUT/fidelity.jsonl --arms "default:" 2>&1 | tee $OUT/fidelity_headline.txt | cut -c1-200
echo "== (2) exchange: saturation threshold"
timeout 900 python tests/experiments/exchange_matrix.py --positions 256 --replicas 2,4 --tiers 8:16 --sat 0,256,2048 --modes 2,0 > $OUT/exchange_matrix.txt 2>$OUT/exchange_matrix.err
grep XM $OUT/exchange_matrix.txt | cut -c1-160; tail -2 $OUT/exchange_matrix.err
echo "== (3) throughput, final policy"
timeout 600 python tests/experiments/arm_bench.py --tokens 60000000 --rounds 2 --arms "default:;noloss:loss=0;resident:window_cache=1;w512:workers=512;w256:workers=256" 2>&1 | grep -v amdgpu.ids | tee $OUT/arm_bench.txt
echo "== done"
