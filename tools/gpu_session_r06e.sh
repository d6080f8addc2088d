#!/bin/bash
# round 6, session e: the quantization-cell rule IN THE LIBRARY (mode 2's default): arithmetic tests, the 2 / 4-replica gates on the
# text8-sized corpus, the 8-replica gates (22 M-token proxy at 131 K words, literal stream at 1 M words), repeats for the spread;
# the changed / new fidelity tests (command line on the literal stream, heldout_v1m, long streams at short rows); bench default.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r06e
mkdir -p $OUT
( python - <<'PY'
import sys
sys.path.insert(0, "tests")
from w2b_testlib import write_headline_corpus
write_headline_corpus("/tmp/headline.txt"); write_headline_corpus("/tmp/cfg1_100m.txt", n_zipf=98_000_000)
PY
) &
CORP=$!
echo "== (1) exchange tests"
timeout 1500 python -m pytest tests/test_gpu_exchange.py -q -m gpu -s 2>&1 | grep -E "EXCHANGE|passed|failed|Error|assert" | tee $OUT/pytest_exchange.txt
wait $CORP
echo "== (2) the library's rule through the sweep harness (lib2 = w2b_exchange_counts + the library's apply): repeats"
RR="timeout 1500 python tests/experiments/replica_rules.py"
$RR /tmp/headline.txt --positions 1024 --rules "lib2;lib2;lib2;signsafe:64:64:0:0:1.0" --out $OUT/rules_p1024.json 2>&1 | grep RR | tee $OUT/rules_p1024.txt
S1=$(python -c "import json; d=json.load(open('$OUT/rules_p1024.json')); print(d['single_replica_loss'], '--single-validation', d['single_replica_validation'])")
$RR /tmp/headline.txt --positions 512 --single $S1 --rules "lib2;lib2" --out $OUT/rules_p512.json 2>&1 | grep RR | tee $OUT/rules_p512.txt
$RR /tmp/headline.txt --positions 2048 --single $S1 --rules "lib2;lib2" --out $OUT/rules_p2048.json 2>&1 | grep RR | tee $OUT/rules_p2048.txt
$RR /tmp/headline.txt --positions 512 --sync geom:1:16 --single $S1 --rules "lib2;lib2" --out $OUT/rules_geom512.json 2>&1 | grep RR | tee $OUT/rules_geom512.txt
$RR /tmp/cfg1_100m.txt --positions 8192 --rules "lib2;lib2" --out $OUT/rules_100m_p8192.json 2>&1 | grep RR | tee $OUT/rules_100m_p8192.txt
S2=$(python -c "import json; d=json.load(open('$OUT/rules_100m_p8192.json')); print(d['single_replica_loss'], '--single-validation', d['single_replica_validation'])")
$RR /tmp/cfg1_100m.txt --positions 4096 --single $S2 --rules "lib2" --out $OUT/rules_100m_p4096.json 2>&1 | grep RR | tee $OUT/rules_100m_p4096.txt
$RR /tmp/cfg1_100m.txt --positions 2048 --single $S2 --rules "lib2" --out $OUT/rules_100m_p2048.json 2>&1 | grep RR | tee $OUT/rules_100m_p2048.txt
$RR /tmp/cfg1_100m.txt --replicas 2 --positions 8192 --single $S2 --rules "lib2" --out $OUT/rules_100m_r2.json 2>&1 | grep RR | tee $OUT/rules_100m_r2.txt
$RR /tmp/cfg1_100m.txt --replicas 4 --positions 8192 --single $S2 --rules "lib2" --out $OUT/rules_100m_r4.json 2>&1 | grep RR | tee $OUT/rules_100m_r4.txt
rm -f /tmp/cfg1_100m.txt /tmp/headline.txt
echo "== (3) fidelity tests that changed / are new"
timeout 1800 python -m pytest tests/test_gpu_fidelity.py -q -m gpu -s -k "literally or second_held_out or long_streams" 2>&1 | grep -E "FIDELITY|passed|failed|Error|assert" | cut -c1-400 | tee $OUT/pytest_fidelity_new.txt
echo "== (4) bench (the driver's command)"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.log 2>$OUT/bench_default.err
tail -1 $OUT/bench_default.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('BENCH %.2f M words/s, %.3f ms/step, roofline %.4f' % (d['value']/1e6, d['ms_per_step'], r['frac']))
for k,v in d.get('other_shapes',{}).items():
    if isinstance(v,dict) and 'value' in v: print('   ', k, '%.2f M' % (v['value']/1e6), v.get('roofline',{}).get('frac'))
"
echo "== done"
