#!/bin/bash
# round 6, session b:
#  host (background): reference bands of the two LONG-stream short-row regimes (long_d200, long_d400b2: 100 M tokens, the row
#        lengths at which the row-group kernel is automatic), on one corpus file that ./word2bits then trains on as well;
#  GPU:  (1) what the combination of 8 replicas' deltas SHOULD be: per-row least-squares factor on the sum against a truth run
#        (tests/experiments/replica_truth.py), and the replicas' epoch loss when they adopt the truth at every exchange (what
#        the interval alone costs) at 16 K / 131 K / 1 M words per replica; (2) counters for the store policy of the per-XCD
#        copies (nt against write-back: same time in session a -- do the writes leave the fabric at all?); (3) the new held-out
#        full-device regime heldout_v1m under the shipped defaults; (4) the cfg5-shape legs, round-4 library against HEAD on ONE
#        box (the round-5 review: cfg5 fell on the driver's box while the headline rose); (5) the new exchange tests.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r06b
mkdir -p $OUT /tmp/w2b_bands_a
R=$GRAFT_REPO_ROOT
python - <<'PY'
import sys, time
sys.path.insert(0, "tests")
from w2b_testlib import write_heldout_corpus, write_headline_corpus
t = time.time(); write_heldout_corpus("/tmp/long.txt", "long_d200"); print("long corpus written in %.0f s" % (time.time() - t), flush=True)
PY
( python tests/golden/make_fidelity_bands.py --out $OUT/bands_long_d200.json --jobs long_d200 --heldout-big 256x1 --reuse-corpus /tmp/long.txt --tmp /tmp/w2b_bands_a > $OUT/bands_long_d200.log 2>&1;
  python tests/golden/make_fidelity_bands.py --out $OUT/bands_long_d400b2.json --jobs long_d400b2 --heldout-big 256x1 --reuse-corpus /tmp/long.txt --tmp /tmp/w2b_bands_a > $OUT/bands_long_d400b2.log 2>&1;
  echo "reference bands done" ) &
REF=$!
python - <<'PY'
import sys, time
sys.path.insert(0, "tests")
from w2b_testlib import write_heldout_corpus, write_headline_corpus
t = time.time(); write_headline_corpus("/tmp/headline.txt"); print("headline corpus written in %.0f s" % (time.time() - t), flush=True)
PY
echo "== (1) truth curves / oracle exchange, 8 replicas x 128 workers, 22 M-token proxy"
RT="timeout 900 python tests/experiments/replica_truth.py /tmp/headline.txt"
$RT --positions 1024 --apply oracle --out $OUT/truth_p1024.json 2>&1 | grep RT | tee $OUT/truth_p1024.txt
$RT --positions 8192 --apply oracle --out $OUT/truth_p8192.json 2>&1 | grep RT | tee $OUT/truth_p8192.txt
$RT --positions 128 --apply oracle --log-at 1,2,4,8,16,32,64,128,160 --out $OUT/truth_p128.json 2>&1 | grep RT | tee $OUT/truth_p128.txt
$RT --positions 1024 --apply smooth:64:64 --out $OUT/truth_p1024_smooth64.json 2>&1 | grep RT | tee $OUT/truth_p1024_smooth64.txt
echo "== (5) exchange tests under the new combination rule"
timeout 900 python -m pytest tests/test_gpu_exchange.py -q -m gpu -x -k "not eight_replicas and not training_effect" 2>&1 | tail -4 | tee $OUT/pytest_exchange.txt
echo "== (2) counters: store policy of the per-XCD copies, nt (A) vs plain write-back (B)"
CMD="python $R/bench.py --steps 8 --warmup 2 --cpu-baseline none --cpu-cfg0 0 --also-relaxed 0 --also-legs 0 --also-shapes 0"
for v in A B; do
  if [ $v = A ]; then L=$R/word2bits_amd/libword2bits_hip.so; else L=$R/word2bits_amd/libword2bits_hip_wb.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && W2B_LIB=$L timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/prof_${v}_$c -o r06 -- $CMD > $R/$OUT/rocprof_${v}_$c.log 2>&1)
  done
  (cd /tmp && W2B_LIB=$L timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$OUT/prof_${v}_l2 -o r06 -- $CMD > $R/$OUT/rocprof_${v}_l2.log 2>&1)
  grep "^{" $OUT/rocprof_${v}_FETCH_SIZE.log | tail -1 > $OUT/bench_pmc_$v.json
  python tools/pmc_summary.py $OUT/prof_${v}_FETCH_SIZE/r06_counter_collection.csv $OUT/prof_${v}_WRITE_SIZE/r06_counter_collection.csv $OUT/prof_${v}_l2/r06_counter_collection.csv $OUT/pmc_store_$v.json k_train_workers $OUT/bench_pmc_$v.json | cut -c1-700
done
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
echo "== (3) heldout_v1m under the shipped defaults"
python - <<'PY'
import sys, time
sys.path.insert(0, "tests")
from w2b_testlib import write_heldout_corpus
t = time.time(); write_heldout_corpus("/tmp/v1m.txt", "heldout_v1m"); print("v1m corpus written in %.0f s" % (time.time() - t), flush=True)
PY
run() {   # tag, file, ref loss, flags
  T0=$(date +%s.%N)
  ./word2bits -train $2 -output /dev/null $4 > $OUT/run.txt 2> $OUT/run.err
  python - "$1" "$3" "$4" <<PY | tee -a $OUT/fidelity_runs.txt
import re, sys
out = open("$OUT/run.txt").read().replace("\r", "\n")
L = [float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", out)]
w = re.search(r"Hogwild workers \(workgroups\): (\d+)", out)
ref = float(sys.argv[2])
err = open("$OUT/run.err").read().strip().replace("\n", " | ")[:160]
print("FID %-12s [%s] workers %s loss %s deviation %s %%  %s" % (sys.argv[1], sys.argv[3][-60:], w.group(1) if w else "?", L, ["%+.2f" % (100 * (x - ref) / abs(ref)) for x in L] if ref else "-", err))
PY
}
FV="-bitlevel 1 -size 512 -window 5 -negative 10 -iter 1 -sample 0 -min-count 5 -binary 1"
for arm in "-threads 0" "-threads 0" "-threads 0 -hot-rows 0" "-threads 256"; do run heldout_v1m /tmp/v1m.txt -340579711.263696 "$FV $arm"; done
rm -f /tmp/v1m.txt
echo "== (4) cfg5 shape: round-4 library vs HEAD on this box"
C5="--vocab 3700000 --dim 1000 --negative 12 --tokens 60000000 --steps 12 --warmup 3 --cpu-baseline none --also-relaxed 0 --also-legs 0 --also-shapes 0"
one() {   # tag, dir, extra flags
  (cd $2 && timeout 600 python bench.py $C5 $3 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('CFG5 %-28s %.2f M words/s, %.3f ms/launch, roofline %.4f, kernel %s, workers %s' % ('$1', d['value']/1e6, r['avg_launch_ms'], r['frac'], r.get('kernel'), d['config'].get('workers')))") | tee -a $OUT/cfg5_ab.txt
}
one "r4 b1 auto" $R/ab_r4 "--cpu-cfg0 0"
one "HEAD b1 auto" $R "--cpu-cfg0 0"
one "HEAD b1 auto period32" $R "--cpu-cfg0 0 --hot-period 32"
one "r4 b0 auto" $R/ab_r4 "--cpu-cfg0 0 --bitlevel 0"
one "HEAD b0 auto" $R "--cpu-cfg0 0 --bitlevel 0"
one "r4 b1 resident" $R/ab_r4 "--cpu-cfg0 0 --window-cache 1"
one "HEAD b1 resident" $R "--cpu-cfg0 0 --window-cache 1"
one "r4 b1 auto" $R/ab_r4 "--cpu-cfg0 0"
one "HEAD b1 auto" $R "--cpu-cfg0 0"
one "HEAD b1 resident" $R "--cpu-cfg0 0 --window-cache 1"
one "r4 b1 resident" $R/ab_r4 "--cpu-cfg0 0 --window-cache 1"
echo "== (6) long streams at short rows: ./word2bits on the reference's file (the bands are still being recorded)"
FL="-bitlevel 1 -size 200 -window 8 -negative 24 -iter 1 -min-count 5 -binary 1"
for arm in "-threads 0" "-threads 256" "-threads 256 -row-groups 0" "-threads 64"; do run long_d200 /tmp/long.txt 0 "$FL $arm"; done
FL="-bitlevel 2 -size 400 -window 8 -negative 24 -iter 1 -min-count 5 -binary 1"
for arm in "-threads 0" "-threads 256" "-threads 256 -row-groups 0" "-threads 64"; do run long_d400b2 /tmp/long.txt 0 "$FL $arm"; done
echo "== waiting for the host's reference runs"
wait $REF
rm -f /tmp/long.txt /tmp/headline.txt
tail -2 $OUT/bands_long_d200.log $OUT/bands_long_d400b2.log
echo "== done"
