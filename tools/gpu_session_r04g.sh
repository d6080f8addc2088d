#!/bin/bash
# round 4, session g: per-XCD copies as READ CACHES (hot_mode 2): every update goes to the master row (u: atomic add, v: store)
# and to the copy, merges are refreshes -- throughput and fidelity at every worker count and regime.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r04g
mkdir -p $OUT
echo "== (0) bit identity"
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_worker.py -k "round4_knob or resident_kernel_equals" 2>&1 | tail -3
echo "== (1) throughput, headline shape"
timeout 900 python tests/experiments/arm_bench.py --tokens 60000000 --rounds 2 --out $OUT/arm_bench.json --arms "default:;cache:hot_mode=2;cache_p8:hot_mode=2,hot_period=8;cache_p2:hot_mode=2,hot_period=2;cache_cap32:hot_mode=2,hot_cap=32;cache_w512:hot_mode=2,workers=512;cache_w256:hot_mode=2,workers=256;w512:workers=512" 2>&1 | grep -v amdgpu.ids | tee $OUT/arm_bench.txt
echo "== (1b) throughput, size 200"
timeout 600 python tests/experiments/arm_bench.py --dim 200 --vocab 60238 --tokens 60000000 --rounds 1 --arms "d200 default:;d200 cache:hot_mode=2;d200 resident:window_cache=1" 2>&1 | grep -v amdgpu.ids | tee $OUT/arm_bench_d200.txt
echo "== (2) fidelity, benchmarked regime"
timeout 1200 python tests/experiments/fidelity_matrix.py --jobs headline --threads 64,256,0,1024 --kernel auto --out $OUT/fidelity.jsonl --arms "cache:-hot-mode 2;cache_p8:-hot-mode 2 -hot-period 8;cache_p2:-hot-mode 2 -hot-period 2" 2>&1 | tee $OUT/fidelity_headline.txt | cut -c1-200
echo "== (3) the other regimes"
timeout 900 python tests/experiments/fidelity_matrix.py --jobs heldout_zipf12,heldout_k5,text8size --threads 0,256 --kernel auto --out $OUT/fidelity.jsonl --arms "cache:-hot-mode 2" 2>&1 | tee $OUT/fidelity_other.txt | cut -c1-200
echo "== done"
