#!/bin/bash
set +e
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_eval.py tests/test_gpu_worker.py tests/test_gpu_cli.py tests/test_gpu_integration.py "tests/test_gpu_fidelity.py::test_cfg2_shape_2bit_d400_accuracy_parity" -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | grep -E "^E  |^tests/|passed|failed|Error" | cut -c1-300
