#!/bin/bash
# round 6, session c: (1) the row-group kernel after the error sum was split over all data wavefronts: parity suite (bit-identity
# with the plain kernel, oracle tests), throughput of the two short-row legs, and the s_waitcnt drain before the end-of-word
# barrier as an A/B (W2B_GROUPS_DRAIN 0 / 1 / 2); (2) the MEASURED combination curve (tests/experiments/replica_curve_r06b.json,
# from session b's truth runs) as the exchange rule of 8 replicas: uniform and doubling intervals, 22 M-token proxy and the literal
# 100 M-token stream, epoch loss AND the final model's loss on a fixed sample; (3) the command line on the literal stream with an
# explicit -threads 256 (now: the library's own choice, with a notice); (4) cfg5 shape, sentence-resident kernel, merge period 32.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r06c
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
( python - <<'PY'
import sys, time
sys.path.insert(0, "tests")
from w2b_testlib import write_headline_corpus
t = time.time(); write_headline_corpus("/tmp/headline.txt"); print("headline corpus written in %.0f s" % (time.time() - t), flush=True)
t = time.time(); write_headline_corpus("/tmp/cfg1_100m.txt", n_zipf=98_000_000); print("cfg1 corpus written in %.0f s" % (time.time() - t), flush=True)
PY
) > $OUT/corpora.log 2>&1 &
CORP=$!
echo "== (1) row-group kernel: parity suite"
timeout 900 python -m pytest tests/test_gpu_groups.py -q -m gpu -x 2>&1 | tail -4 | tee $OUT/pytest_groups.txt
S="--steps 12 --warmup 3 --tokens 30000000 --cpu-baseline none --cpu-cfg0 0 --also-relaxed 0 --also-legs 0 --also-shapes 0"
leg() {   # tag, lib, flags
  W2B_LIB=$2 timeout 300 python bench.py $S $3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('GROUPS %-22s %.2f M words/s, %.3f ms/launch, kernel %s, workers %s' % ('$1', d['value']/1e6, r['avg_launch_ms'], r.get('kernel'), d['config'].get('workers')))" | tee -a $OUT/groups_bench.txt
}
for i in 1 2; do
  for v in "" _drain1 _drain2; do
    leg "d200 drain${v:-0}" $R/word2bits_amd/libword2bits_hip$v.so "--vocab 60238 --dim 200"
    leg "d400b2 drain${v:-0}" $R/word2bits_amd/libword2bits_hip$v.so "--vocab 60238 --dim 400 --bitlevel 2"
  done
done
W2B_LIB=$R/word2bits_amd/libword2bits_hip_drain2.so timeout 600 python -m pytest tests/test_gpu_groups.py -q -m gpu -x -k "equals_plain" 2>&1 | tail -2 | tee $OUT/pytest_groups_drain2.txt
wait $CORP; cat $OUT/corpora.log
echo "== (2) the measured curve as the exchange rule: 22 M-token proxy"
RR="timeout 1500 python tests/experiments/replica_rules.py"
C=tests/experiments/replica_curve_r06b.json
$RR /tmp/headline.txt --positions 1024 --rules "table:$C;table:$C:0.8;table:$C:1.2;lib2" --out $OUT/rules_p1024.json 2>&1 | grep RR | tee $OUT/rules_p1024.txt
S1=$(python -c "import json; d=json.load(open('$OUT/rules_p1024.json')); print(d['single_replica_loss'], '--single-validation', d['single_replica_validation'])")
$RR /tmp/headline.txt --positions 128 --single $S1 --rules "table:$C" --out $OUT/rules_p128.json 2>&1 | grep RR | tee $OUT/rules_p128.txt
$RR /tmp/headline.txt --positions 128 --sync geom:1:64 --single $S1 --rules "table:$C" --out $OUT/rules_geom128.json 2>&1 | grep RR | tee $OUT/rules_geom128.txt
$RR /tmp/headline.txt --positions 256 --sync geom:1:32 --single $S1 --rules "table:$C" --out $OUT/rules_geom256.json 2>&1 | grep RR | tee $OUT/rules_geom256.txt
$RR /tmp/headline.txt --positions 512 --sync geom:1:16 --single $S1 --rules "table:$C" --out $OUT/rules_geom512.json 2>&1 | grep RR | tee $OUT/rules_geom512.txt
$RR /tmp/headline.txt --positions 8192 --single $S1 --rules "table:$C" --out $OUT/rules_p8192.json 2>&1 | grep RR | tee $OUT/rules_p8192.txt
echo "== (2b) the literal 100 M-token stream"
$RR /tmp/cfg1_100m.txt --positions 1024 --rules "table:$C" --out $OUT/rules_100m_p1024.json 2>&1 | grep RR | tee $OUT/rules_100m_p1024.txt
S2=$(python -c "import json; d=json.load(open('$OUT/rules_100m_p1024.json')); print(d['single_replica_loss'], '--single-validation', d['single_replica_validation'])")
$RR /tmp/cfg1_100m.txt --positions 2048 --single $S2 --rules "table:$C" --out $OUT/rules_100m_p2048.json 2>&1 | grep RR | tee $OUT/rules_100m_p2048.txt
$RR /tmp/cfg1_100m.txt --positions 8192 --single $S2 --rules "table:$C" --out $OUT/rules_100m_p8192.json 2>&1 | grep RR | tee $OUT/rules_100m_p8192.txt
$RR /tmp/cfg1_100m.txt --positions 256 --sync geom:1:32 --single $S2 --rules "table:$C" --out $OUT/rules_100m_geom256.json 2>&1 | grep RR | tee $OUT/rules_100m_geom256.txt
$RR /tmp/cfg1_100m.txt --positions 512 --sync geom:1:16 --single $S2 --rules "table:$C" --out $OUT/rules_100m_geom512.json 2>&1 | grep RR | tee $OUT/rules_100m_geom512.txt
echo "== (3) the command line, literal stream, explicit -threads 256"
F="-bitlevel 1 -size 800 -window 8 -negative 24 -iter 1 -sample 0 -min-count 5 -binary 1"
for arm in "-threads 256" "-threads 256 -threads-literal 1"; do
  ./word2bits -train /tmp/cfg1_100m.txt -output /dev/null $F $arm > $OUT/run.txt 2> $OUT/run.err
  echo "CLI [$arm]: $(grep -o 'Hogwild workers (workgroups): [0-9]*' $OUT/run.txt) $(tr '\r' '\n' < $OUT/run.txt | grep 'Epoch Loss') | $(head -c 200 $OUT/run.err)" | tee -a $OUT/cli_literal.txt
done
rm -f /tmp/cfg1_100m.txt /tmp/headline.txt
echo "== (4) cfg5 shape, sentence-resident kernel: merge period 16 (HEAD) against 32 (round 4's)"
C5="--vocab 3700000 --dim 1000 --negative 12 --tokens 60000000 --steps 12 --warmup 3 --cpu-baseline none --cpu-cfg0 0 --also-relaxed 0 --also-legs 0 --also-shapes 0 --window-cache 1"
for arm in "" "--hot-period 32" "" "--hot-period 32"; do
  timeout 600 python bench.py $C5 $arm 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('CFG5 resident [%s] %.2f M words/s, %.3f ms/launch, roofline %.4f' % ('$arm', d['value']/1e6, r['avg_launch_ms'], r['frac']))" | tee -a $OUT/cfg5_resident_period.txt
done
echo "== done"
