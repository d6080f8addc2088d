#!/bin/bash
# round 2, session N: the whole GPU suite on the consolidated kernel + fidelity calibration output
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out/r02n
timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short --durations=12 -s 2>&1 | grep -E "^E  |^tests/|passed|failed|Error|s call|FIDELITY" | cut -c1-330 | tee gpurun_out/r02n/pytest.txt
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-52s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
timeout 900 python bench.py 2>gpurun_out/r02n/bench.err | tee gpurun_out/r02n/bench_default.json | cut -c1-1500
echo "== done"
