#!/bin/bash
# round 6, session j: the fidelity gates after the planted-corpus exception was retired (3/8 of the workers at a time) and with the
# two-run bands of heldout_v1m / long_d200 / long_d400b2
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r06j
mkdir -p $OUT
timeout 2400 python -m pytest tests/test_gpu_fidelity.py -q -m gpu -s 2>&1 | grep -E "FIDELITY|passed|failed|Error|assert" | cut -c1-330 | tee $OUT/pytest_fidelity.txt
echo "== done"
