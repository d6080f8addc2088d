#!/bin/bash
set +e
OUT=gpurun_out/s2
mkdir -p $OUT
export TMPDIR=/tmp
echo "== coherence2"; timeout 300 ./tools/coherence_probe2 > $OUT/coherence2.log 2>&1; cat $OUT/coherence2.log
echo "== worker diag"; timeout 600 python tools/worker_diag.py > $OUT/worker_diag.log 2>&1; cat $OUT/worker_diag.log | tail -60
B="python bench.py --tokens 20000000 --steps 8 --warmup 2 --cpu-baseline none"
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-28s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
for v in "" _t9w4 _t25w2 _nostash _sc1; do
  W2B_LIB=$PWD/word2bits_amd/libword2bits_hip$v.so timeout 600 $B 2>$OUT/err_t$v.log | short "tuples$v" | tee -a $OUT/variants.log
done
for v in "" _sc1 _t9w4; do
  W2B_LIB=$PWD/word2bits_amd/libword2bits_hip$v.so timeout 600 $B --form worker 2>$OUT/err_w$v.log | short "worker$v" | tee -a $OUT/variants.log
done
for g in 1 2 3 4 6; do
  W2B_GRID_PER_CU=$g timeout 600 $B 2>>$OUT/err_g.log | short "tuples grid_per_cu=$g" | tee -a $OUT/variants.log
done
for w in 512 768 1536 2048; do
  timeout 600 $B --form worker --workers $w --positions $((1048576/w)) 2>>$OUT/err_g.log | short "worker workers=$w" | tee -a $OUT/variants.log
done
timeout 600 $B --ids uniform 2>>$OUT/err_g.log | short "tuples uniform" | tee -a $OUT/variants.log
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -12 | tee $OUT/pytest.log
echo "== done"
