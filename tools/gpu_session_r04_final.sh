#!/bin/bash
# round 4, final session: (1) the whole -m gpu suite; (2) the driver's bench command; (3) rocprofv3 kernel stats and the two
# PMC passes of the same command (calibrated counters: profiles/r04_pmc_calibration.json); (4) cache-resident bound for the
# short-row legs; (5) end-to-end wall time of ./word2bits and of the reference program on one file (the reference runs on
# the host cores while the GPU profiles); (6) the evaluator.   RND / SKIP_TESTS can be set from outside.
set +e
RND=${RND:-r04}
OUT=gpurun_out/${RND}_final
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ -z "$SKIP_TESTS" ]; then
echo "== (1) pytest -m gpu"
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
fi
echo "== (5a) end to end: ./word2bits"
timeout 600 python tools/e2e_compare.py $OUT/e2e.json --only hip 2>&1 | tail -1 | cut -c1-400
echo "== (2) bench (the driver's command)"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.log 2>$OUT/bench_default.err
tail -1 $OUT/bench_default.log | cut -c1-1200
echo "== (5b) end to end: the reference program on the host cores (background)"
(timeout 1500 python tools/e2e_compare.py $OUT/e2e.json --only ref > $OUT/e2e_ref.log 2>&1; echo e2e-ref done) &
E2E=$!
K=$(tail -1 $OUT/bench_default.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['roofline']['kernel'])")
CMD="python $R/bench.py --steps 8 --warmup 2 --cpu-baseline none --also-relaxed 0 --also-legs 0 --also-shapes 0"
echo "== (3) rocprofv3 of: $CMD   (kernel $K)"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats -o $RND -- $CMD > $R/$OUT/rocprof_stats.log 2>&1)
grep "^{" $OUT/rocprof_stats.log | tail -1 > $OUT/bench_profiled.json
head -4 $OUT/prof_stats/${RND}_kernel_stats.csv | cut -c1-250
for c in FETCH_SIZE WRITE_SIZE; do
(cd /tmp && timeout 900 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/prof_$c -o $RND -- $CMD > $R/$OUT/rocprof_$c.log 2>&1)
done
(cd /tmp && timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$OUT/prof_l2 -o $RND -- $CMD > $R/$OUT/rocprof_l2.log 2>&1)
grep "^{" $OUT/rocprof_FETCH_SIZE.log | tail -1 > $OUT/bench_pmc.json
python tools/pmc_summary.py $OUT/prof_FETCH_SIZE/${RND}_counter_collection.csv $OUT/prof_WRITE_SIZE/${RND}_counter_collection.csv $OUT/prof_l2/${RND}_counter_collection.csv $OUT/pmc_worker.json $K $OUT/bench_pmc.json | cut -c1-900
echo "== (4) cache-resident bound (row_probe small)"
timeout 120 tools/row_probe small 2>&1 | tee $OUT/row_probe_small.txt
echo "== (6) evaluator"
timeout 300 python bench.py --form eval --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/eval_bench.json; cut -c1-300 $OUT/eval_bench.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_eval -o $RND -- python $R/bench.py --form eval --steps 5 --warmup 1 --eval-cpu-questions 0 > /dev/null 2>&1)
head -3 $OUT/prof_eval/${RND}_kernel_stats.csv | cut -c1-200
echo "== kernel stats of the other shapes (plain = automatic; sentence-resident = explicit)"
S="--steps 8 --warmup 2 --tokens 30000000 --cpu-baseline none --also-relaxed 0 --also-legs 0 --also-shapes 0"
prof() { name=$1; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats_$name -o $RND -- python $R/bench.py $S "$@" > $R/$OUT/rocprof_stats_$name.log 2>&1); tail -1 $OUT/rocprof_stats_$name.log | cut -c1-160; head -2 $OUT/prof_stats_$name/${RND}_kernel_stats.csv | tail -1 | cut -c1-200; }
prof cfg5 --vocab 3700000 --dim 1000 --negative 12
prof d200 --vocab 60238 --dim 200
prof tuples --form tuples
wait $E2E
cat $OUT/e2e.json | tr '\n' ' ' | cut -c1-900; echo
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
echo "== done"
