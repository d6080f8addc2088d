#!/bin/bash
# round 3, session a: reference bands on the box's HOST cores (background) while the GPU runs the test-suite and a
# bench matrix of the XCD-shared hot rows.  Everything lands in gpurun_out/r03a/.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r03a
mkdir -p $OUT
nproc > $OUT/host.txt; lscpu | grep -E "Model name|Socket|Thread|Core" >> $OUT/host.txt
(python tests/golden/make_fidelity_bands.py --out $OUT/bands.json --jobs planted,text8size,headline \
    --planted 8x3,64x3,512x3 --text8size 64x3,256x2 --headline 64x3,256x2 > $OUT/bands.log 2>&1; echo done > $OUT/bands.done) &
BANDS=$!
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short -s 2>&1 > $OUT/pytest.txt
grep -E "FIDELITY|TUPLE HOT|passed|failed|^E  |Error" $OUT/pytest.txt | cut -c1-330 | tail -60
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); wk=d['config'].get('worker_kernel') or {}
        print('%-50s %8.2f Mw/s  frac %.3f  launch %.2f ms  hot %s workers %s' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms'], wk.get('hot_rows_with_xcd_copies'), d['config'].get('workers')))
" "$1"; }
B="python bench.py --cpu-baseline none --also-relaxed 0 --also-legs 0 --tokens 30000000 --steps 10 --warmup 3"
run() { name="$1"; shift; timeout 600 $B "$@" 2>$OUT/err.txt | tee -a $OUT/bench_lines.jsonl | short "$name"; }
echo "== bench matrix"
run "cfg2 default (auto hot rows)"
run "cfg2 hot 0" --hot-rows 0
run "cfg2 hot 16" --hot-rows 16
run "cfg2 hot 64" --hot-rows 64
run "cfg2 hot auto period 2" --hot-period 2
run "cfg2 uniform ids" --ids uniform
run "d200 V60K auto" --dim 200 --vocab 60238
run "d200 V60K hot 0" --dim 200 --vocab 60238 --hot-rows 0
run "d200 V60K hot 128" --dim 200 --vocab 60238 --hot-rows 128
run "d200 V60K uniform" --dim 200 --vocab 60238 --ids uniform
run "d400 b2 V60K auto" --dim 400 --vocab 60238 --bitlevel 2
run "d400 b2 V60K uniform" --dim 400 --vocab 60238 --bitlevel 2 --ids uniform
run "cfg5 b1" --vocab 3700000 --dim 1000 --negative 12
run "tuples auto" --form tuples
run "tuples hot 0" --form tuples --hot-rows 0
run "tuples hot 64" --form tuples --hot-rows 64
run "plain kernel coherent auto" --window-cache 0
echo "== waiting for the reference bands"
for i in $(seq 1 120); do [ -f $OUT/bands.done ] && break; sleep 10; done
tail -3 $OUT/bands.log | cut -c1-300
echo "== done"
