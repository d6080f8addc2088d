#!/bin/bash
set +e
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -k "hot" -m gpu -q --no-header -p no:cacheprovider --tb=short -s 2>&1 | tail -20 | cut -c1-300
