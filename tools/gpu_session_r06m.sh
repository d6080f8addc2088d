#!/bin/bash
# round 6, session m: the -m gpu suite once more on the final tree (are the Hogwild gates stable from box to box?)
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out/r06m
timeout 2400 python -m pytest tests -q -m gpu -s 2>&1 | grep -E "FIDELITY|EXCHANGE|passed|failed|FAILED|Error" | cut -c1-300 | tee gpurun_out/r06m/pytest_gpu.txt
