#!/bin/bash
# round 6, closing session: the whole -m gpu suite on the final tree, then the profile session (tools/gpu_session_r06_final.sh)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06k
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee gpurun_out/r06k/pytest_gpu.txt
bash tools/gpu_session_r06_final.sh
