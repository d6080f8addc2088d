#!/bin/bash
# round 3, session f: exchange mode 2 (sqrt damping of saturated rows), fidelity gates after the rate / policy changes
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r03f
mkdir -p $OUT
echo "== pytest"
timeout 1500 python -m pytest tests/test_gpu_fidelity.py tests/test_gpu_exchange.py -m gpu -q --no-header -p no:cacheprovider --tb=short -s 2>&1 > $OUT/pytest.txt
grep -E "FIDELITY|EXCHANGE|passed|failed|^E  |Error|^FAILED" $OUT/pytest.txt | cut -c1-420 | tail -70
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); wk=d['config'].get('worker_kernel') or {}
        print('%-44s %8.2f Mw/s  frac %.3f  launch %.2f ms  %s hot %s workers %s' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'], wk.get('hot_rows_with_xcd_copies'), d['config'].get('workers')))
" "$1"; }
B="python bench.py --cpu-baseline none --also-relaxed 0 --also-legs 0 --also-shapes 0 --tokens 30000000 --steps 10 --warmup 3"
run() { name="$1"; shift; timeout 600 $B "$@" 2>$OUT/err.txt | tee -a $OUT/bench_lines.jsonl | short "$name"; }
echo "== bench"
run "cfg2 auto"
run "d200 V60K auto" --dim 200 --vocab 60238
run "d400 b2 V60K auto" --dim 400 --vocab 60238 --bitlevel 2
run "cfg5 b1 auto" --vocab 3700000 --dim 1000 --negative 12
echo "== done"
