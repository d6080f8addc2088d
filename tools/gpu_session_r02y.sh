#!/bin/bash
set +e
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_replicas.py -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -8 | cut -c1-400
