#!/bin/bash
# last session of a round: the whole -m gpu suite, then the profile session
set +e
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -15 | cut -c1-300
bash tools/gpu_profile_session.sh
