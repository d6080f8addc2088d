#!/bin/bash
# round 3, session e: the new fidelity gates, exchange, wide rows, evaluator cross-check + the bench matrix of the
# kernel-choice policy
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r03e
mkdir -p $OUT
echo "== pytest"
timeout 1500 python -m pytest tests/test_gpu_fidelity.py tests/test_gpu_exchange.py tests/test_gpu_exact.py tests/test_gpu_eval.py tests/test_gpu_parity.py tests/test_gpu_worker.py -m gpu -q --no-header -p no:cacheprovider --tb=short -s 2>&1 > $OUT/pytest.txt
grep -E "FIDELITY|EXCHANGE|TUPLE HOT|passed|failed|^E  |Error|^FAILED" $OUT/pytest.txt | cut -c1-400 | tail -70
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
run "cfg2 auto loss on" --loss 1
run "cfg2 resident" --window-cache 1
run "cfg2 plain hot-cap 128" --window-cache 0 --hot-cap 128
run "cfg2 plain hot-period 32" --window-cache 0 --hot-period 32
run "d200 V60K auto" --dim 200 --vocab 60238
run "d200 V60K resident" --dim 200 --vocab 60238 --window-cache 1
run "d400 b2 V60K auto" --dim 400 --vocab 60238 --bitlevel 2
run "d400 b2 V60K resident" --dim 400 --vocab 60238 --bitlevel 2 --window-cache 1
run "cfg5 b1 auto" --vocab 3700000 --dim 1000 --negative 12
run "cfg5 b1 plain" --vocab 3700000 --dim 1000 --negative 12 --window-cache 0
run "cfg5 b0 auto" --vocab 3700000 --dim 1000 --negative 12 --bitlevel 0
run "cfg2 uniform auto" --ids uniform
run "tuples" --form tuples
echo "== done"
