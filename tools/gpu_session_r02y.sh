#!/bin/bash
# last check of the round: smoke(), the worker + fidelity(text8) tests, the default line
set +e
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_worker.py tests/test_gpu_cli.py -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -3 | cut -c1-200
timeout 600 python bench.py --cpu-baseline none 2>&1 | tail -1 | cut -c1-200
