#!/bin/bash
# The round's profile session (run through gpurun): default bench line, rocprofv3 kernel stats and the PMC passes of
# the SAME command, the cfg5-shape line + stats, the evaluator, the tuples form, the matrix-pipe probe.  Summaries are copied to profiles/ by hand afterwards.
set +e
RND=${RND:-r03}
OUT=gpurun_out/${RND}_profile
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== bench default"; timeout 1200 python bench.py > $OUT/bench_default.log 2>$OUT/bench_default.err; tail -1 $OUT/bench_default.log | cut -c1-600
form=worker
K=$(tail -1 $OUT/bench_default.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['roofline']['kernel'])")
echo "kernel of the default workload: $K"
CMD="python $R/bench.py --form $form --steps 8 --warmup 2 --cpu-baseline none --also-relaxed 0 --also-legs 0 --also-shapes 0"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats_$form -o $RND -- $CMD > $R/$OUT/rocprof_stats_$form.log 2>&1)
tail -1 $OUT/rocprof_stats_$form.log > $OUT/bench_profiled_$form.json
(cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/prof_fetch_$form -o $RND -- $CMD > $R/$OUT/rocprof_fetch_$form.log 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/prof_write_$form -o $RND -- $CMD > $R/$OUT/rocprof_write_$form.log 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$OUT/prof_l2_$form -o $RND -- $CMD > $R/$OUT/rocprof_l2_$form.log 2>&1)
python tools/pmc_summary.py $OUT/prof_fetch_$form/${RND}_counter_collection.csv $OUT/prof_write_$form/${RND}_counter_collection.csv $OUT/prof_l2_$form/${RND}_counter_collection.csv $OUT/pmc_$form.json $K $OUT/bench_default.log | cut -c1-700
head -4 $OUT/prof_stats_$form/${RND}_kernel_stats.csv | cut -c1-250
echo "== the other shapes: kernel stats (cfg5 b1, d200, d400 b2, sentence-resident kernel forced on the default workload)"
S="--steps 8 --warmup 2 --tokens 30000000 --cpu-baseline none --also-relaxed 0 --also-legs 0 --also-shapes 0"
prof() { name=$1; shift; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats_$name -o $RND -- python $R/bench.py $S "$@" > $R/$OUT/rocprof_stats_$name.log 2>&1); tail -1 $OUT/rocprof_stats_$name.log | cut -c1-200; head -3 $OUT/prof_stats_$name/${RND}_kernel_stats.csv | cut -c1-250; }
prof cfg5 --vocab 3700000 --dim 1000 --negative 12
prof d200 --vocab 60238 --dim 200
prof d400b2 --vocab 60238 --dim 400 --bitlevel 2
prof resident --window-cache 1
echo "== PMC: cfg5 and tuples"
pmc() { name=$1; ksub=$2; shift; shift; for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout 900 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/prof_${c}_$name -o $RND -- python $R/bench.py $S "$@" > $R/$OUT/rocprof_${c}_$name.log 2>&1); done; tail -1 $OUT/rocprof_FETCH_SIZE_$name.log > $OUT/bench_pmc_$name.json; python tools/pmc_summary.py $OUT/prof_FETCH_SIZE_$name/${RND}_counter_collection.csv $OUT/prof_WRITE_SIZE_$name/${RND}_counter_collection.csv - $OUT/pmc_$name.json $ksub $OUT/bench_pmc_$name.json | cut -c1-500; }
pmc cfg5 k_train_resident --vocab 3700000 --dim 1000 --negative 12
pmc tuples k_train_tuples --form tuples
pmc d200 k_train_ --vocab 60238 --dim 200
echo "== evaluator"
timeout 300 python bench.py --form eval --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/eval_bench.json; cut -c1-300 $OUT/eval_bench.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_eval -o $RND -- python $R/bench.py --form eval --steps 5 --warmup 1 --eval-cpu-questions 0 > /dev/null 2>&1)
head -3 $OUT/prof_eval/${RND}_kernel_stats.csv | cut -c1-200
echo "== tuples form (coherent rows)"
timeout 300 python bench.py --form tuples --cpu-baseline none --also-shapes 0 2>/dev/null | tail -1 > $OUT/bench_tuples.json; cut -c1-300 $OUT/bench_tuples.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats_tuples -o $RND -- python $R/bench.py --form tuples --steps 8 --warmup 2 --cpu-baseline none --also-relaxed 0 --also-shapes 0 > /dev/null 2>&1)
head -3 $OUT/prof_stats_tuples/${RND}_kernel_stats.csv | cut -c1-250
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
echo "== done"
