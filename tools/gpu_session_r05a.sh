#!/bin/bash
# round 5, session a: first contact of the row-group worker kernel (w2b_kernels_groups.hip) with the GPU --
# parity against the plain kernel, then throughput at the short-row shapes and at the headline shape, plain vs groups
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r05a
mkdir -p $OUT
timeout 420 python -m pytest tests/test_gpu_groups.py -x -q -m gpu -s > $OUT/pytest_groups.txt 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest_groups.txt
tail -15 $OUT/pytest_groups.txt
A="plain256:row_groups=0,workers=256,loss=1;groups256:row_groups=1,workers=256,loss=1;groups512:row_groups=1,workers=512,loss=1;plain512:row_groups=0,workers=512,loss=1"
timeout 200 python tests/experiments/arm_bench.py --vocab 60238 --dim 200 --rounds 1 --arms "$A" --out $OUT/arm_d200.json > $OUT/arm_d200.txt 2>&1; tail -6 $OUT/arm_d200.txt
timeout 200 python tests/experiments/arm_bench.py --vocab 60238 --dim 400 --bitlevel 2 --rounds 1 --arms "$A" --out $OUT/arm_d400.json > $OUT/arm_d400.txt 2>&1; tail -6 $OUT/arm_d400.txt
timeout 200 python tests/experiments/arm_bench.py --rounds 1 --arms "plain256:row_groups=0,workers=256,loss=1;groups256:row_groups=1,workers=256,loss=1;plain1024:loss=1" --out $OUT/arm_d800.json > $OUT/arm_d800.txt 2>&1; tail -5 $OUT/arm_d800.txt
echo "== done"
