#!/bin/bash
# round 4, session n: the log-sigmoid terms booked by wavefront 1 at the top of the next position (off wavefront 0's critical
# path): tests that check the loss against the oracle, then this build beside the previous one on this box.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r04n
mkdir -p $OUT
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_parity.py tests/test_gpu_exact.py tests/test_gpu_worker.py tests/test_gpu_integration.py tests/test_gpu_cli.py 2>&1 | tail -3 | tee $OUT/pytest.txt
timeout 600 python tests/experiments/arm_bench.py --tokens 60000000 --rounds 3 --arms "new loss:loss=1;new noloss:loss=0" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/ab_new.txt
W2B_LIB=$PWD/word2bits_amd/libword2bits_hip_prev.so timeout 600 python tests/experiments/arm_bench.py --tokens 60000000 --rounds 3 --arms "prev loss:loss=1;prev noloss:loss=0" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/ab_prev.txt
echo "== done"
