#!/bin/bash
# round 3, session i: hot_cap 128 with hot_period 16 / 32 -- fidelity gates and speed
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r03i
mkdir -p $OUT
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
run "cfg2 default"
for p in 16 32; do
  run "cfg2 period $p cap 128" --hot-period $p --hot-cap 128
  run "d200 period $p cap 128" --dim 200 --vocab 60238 --hot-period $p --hot-cap 128
  run "d400 b2 period $p cap 128" --dim 400 --vocab 60238 --bitlevel 2 --hot-period $p --hot-cap 128
  run "cfg5 b1 period $p cap 128" --vocab 3700000 --dim 1000 --negative 12 --hot-period $p --hot-cap 128
  run "tuples period $p cap 128" --form tuples --hot-period $p --hot-cap 128
done
for p in 16 32; do
echo "== fidelity with -hot-period $p -hot-cap 128"
timeout 900 python tests/experiments/fidelity_matrix.py text8size headline -hot-period=$p -hot-cap=128 2>&1 | tee $OUT/fidelity_p${p}_c128.txt | cut -c1-250
done
echo "== done"
