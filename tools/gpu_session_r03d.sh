#!/bin/bash
# round 3, session d: lossless hot rows (atomic adds + CAS merges) -- exactness tests, the exchange tests, a fidelity
# matrix against the host's reference bands, and the price of atomic updates in the bench.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r03d
mkdir -p $OUT
echo "== pytest (exchange)"
timeout 900 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_parity.py -m gpu -q --no-header -p no:cacheprovider --tb=short -s 2>&1 > $OUT/pytest.txt
grep -E "EXCHANGE|TUPLE HOT|passed|failed|^E  |Error|^FAILED" $OUT/pytest.txt | cut -c1-300 | tail -40
echo "== fidelity matrix"
timeout 1200 python tests/experiments/fidelity_matrix.py text8size headline 2>&1 | tee $OUT/fidelity_matrix.txt | cut -c1-250
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); wk=d['config'].get('worker_kernel') or {}
        print('%-50s %8.2f Mw/s  frac %.3f  launch %.2f ms  hot %s workers %s' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms'], wk.get('hot_rows_with_xcd_copies'), d['config'].get('workers')))
" "$1"; }
B="python bench.py --cpu-baseline none --also-relaxed 0 --also-legs 0 --tokens 30000000 --steps 10 --warmup 3"
run() { name="$1"; shift; timeout 600 $B "$@" 2>$OUT/err.txt | tee -a $OUT/bench_lines.jsonl | short "$name"; }
echo "== bench"
run "cfg2 auto"
run "cfg2 auto loss on" --loss 1
run "cfg2 plain hot-weight 500" --window-cache 0 --hot-weight 500
run "cfg2 plain hot-cap 128" --window-cache 0 --hot-cap 128
run "cfg2 plain hot-period 2" --window-cache 0 --hot-period 2
run "d200 V60K auto" --dim 200 --vocab 60238
run "d200 V60K resident" --dim 200 --vocab 60238 --window-cache 1
run "d400 b2 V60K auto" --dim 400 --vocab 60238 --bitlevel 2
run "d400 b2 V60K resident" --dim 400 --vocab 60238 --bitlevel 2 --window-cache 1
run "cfg5 b1 auto" --vocab 3700000 --dim 1000 --negative 12
run "cfg5 b1 plain" --vocab 3700000 --dim 1000 --negative 12 --window-cache 0
run "cfg5 b0 auto" --vocab 3700000 --dim 1000 --negative 12 --bitlevel 0
run "cfg2 uniform auto" --ids uniform
echo "== done"
