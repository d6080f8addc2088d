#!/bin/bash
# round 4, last session: the whole -m gpu suite on the final tree, the driver's bench command, and the final build beside the
# round-3 library on ONE box (W2B_LIB).
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r04h
mkdir -p $OUT
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
echo "== smoke"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench (the driver's command)"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.log 2>$OUT/bench_default.err
tail -1 $OUT/bench_default.log | cut -c1-700
echo "== A/B on this box: final build vs the round-3 library"
timeout 600 python tests/experiments/arm_bench.py --tokens 60000000 --rounds 2 --arms "r04 loss:loss=1;r04 noloss:loss=0" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $OUT/ab_r04.txt
W2B_LIB_ALLOW_MISSING=1 W2B_LIB=$PWD/word2bits_amd/libword2bits_hip_r03.so timeout 600 python tests/experiments/arm_bench.py --tokens 60000000 --rounds 2 --arms "r03 loss:loss=1;r03 noloss:loss=0" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $OUT/ab_r03.txt
echo "== done"
