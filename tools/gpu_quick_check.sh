#!/bin/bash
# a few GPU minutes: the tests of what changed last (given as arguments, default: packed export + command line), then smoke()
set +e
export TMPDIR=/tmp W2B_TEST_NO_TORCH=1
T="${@:-tests/test_packed_vectors.py tests/test_gpu_cli.py}"
mkdir -p gpurun_out
timeout 200 python -m pytest $T -m gpu -x -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -25 | cut -c1-300 | tee gpurun_out/quick_check.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a gpurun_out/quick_check.txt
