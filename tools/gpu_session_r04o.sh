#!/bin/bash
# round 4, session o: more per-XCD copies on a full device (W2B_XHOT_MAX 512 build): speed and fidelity at 1024 workers.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r04o
mkdir -p $OUT
export W2B_LIB=$PWD/word2bits_amd/libword2bits_hip_x512.so
timeout 600 python tests/experiments/arm_bench.py --tokens 60000000 --rounds 2 --arms "cap128:hot_cap=128;cap256:hot_cap=256;cap512:hot_cap=512;cap256_p64:hot_cap=256,hot_period=64;cap128_p64:hot_period=64" 2>&1 | grep -v amdgpu.ids | tail -7 | tee $OUT/arm_bench.txt
