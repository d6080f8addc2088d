#!/bin/bash
# the whole -m gpu suite at the round's last commit
set +e
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -8 | cut -c1-300
