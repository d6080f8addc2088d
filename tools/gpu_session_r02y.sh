#!/bin/bash
# up to 8 private hot rows (same rate threshold) instead of 4: speed at three shapes and the text8-sized fidelity test
set +e
export TMPDIR=/tmp
run() { timeout 300 python bench.py --cpu-baseline none --also-relaxed 0 --also-legs 0 --steps 12 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cap $W2B_HOT_CAP $* ->', round(d['value']/1e6,2), 'M', round(d['roofline']['frac'],3), d['config']['worker_kernel']['private_hot_rows'])"; }
for cap in 4 8; do export W2B_HOT_CAP=$cap; run; run --vocab 60000 --dim 200; run --vocab 60000 --dim 400 --bitlevel 2; done
export W2B_HOT_CAP=8
timeout 900 python -m pytest "tests/test_gpu_fidelity.py::test_text8_size_threads0_resident_vs_plain_vs_reference" -m gpu -q --no-header -p no:cacheprovider --tb=short -s 2>&1 | grep -E "FIDELITY|passed|failed|^E " | cut -c1-300
