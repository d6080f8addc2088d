#!/bin/bash
# The round's profile session (run through gpurun): default bench line, rocprofv3 kernel stats and the PMC passes of
# the SAME command, the cfg5-shape line + stats, the evaluator, the tuples form, the matrix-pipe probe.  Summaries are copied to profiles/ by hand afterwards.
set +e
RND=${RND:-r02}
OUT=gpurun_out/${RND}_profile
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== bench default"; timeout 1200 python bench.py > $OUT/bench_default.log 2>$OUT/bench_default.err; tail -1 $OUT/bench_default.log | cut -c1-600
form=worker; K=k_train_resident
CMD="python $R/bench.py --form $form --steps 8 --warmup 2 --cpu-baseline none --also-relaxed 0 --also-legs 0"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats_$form -o $RND -- $CMD > $R/$OUT/rocprof_stats_$form.log 2>&1)
tail -1 $OUT/rocprof_stats_$form.log > $OUT/bench_profiled_$form.json
(cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/prof_fetch_$form -o $RND -- $CMD > $R/$OUT/rocprof_fetch_$form.log 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/prof_write_$form -o $RND -- $CMD > $R/$OUT/rocprof_write_$form.log 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$OUT/prof_l2_$form -o $RND -- $CMD > $R/$OUT/rocprof_l2_$form.log 2>&1)
python tools/pmc_summary.py $OUT/prof_fetch_$form/${RND}_counter_collection.csv $OUT/prof_write_$form/${RND}_counter_collection.csv $OUT/prof_l2_$form/${RND}_counter_collection.csv $OUT/pmc_$form.json $K $OUT/bench_default.log | cut -c1-700
head -4 $OUT/prof_stats_$form/${RND}_kernel_stats.csv | cut -c1-250
echo "== cfg5 shape (V=3.7M, D=1000, negative 12), bitlevel 1 and 0"
for b in 1 0; do
  timeout 900 python bench.py --vocab 3700000 --dim 1000 --negative 12 --bitlevel $b --cpu-baseline none --also-legs 0 > $OUT/bench_cfg5_b$b.log 2>/dev/null; tail -1 $OUT/bench_cfg5_b$b.log | cut -c1-400
done
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats_cfg5 -o $RND -- python $R/bench.py --vocab 3700000 --dim 1000 --negative 12 --steps 8 --warmup 2 --cpu-baseline none --also-relaxed 0 --also-legs 0 > $R/$OUT/rocprof_stats_cfg5.log 2>&1)
head -3 $OUT/prof_stats_cfg5/${RND}_kernel_stats.csv | cut -c1-250
echo "== evaluator"
timeout 300 python bench.py --form eval --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/eval_bench.json; cut -c1-300 $OUT/eval_bench.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_eval -o $RND -- python $R/bench.py --form eval --steps 5 --warmup 1 --eval-cpu-questions 0 > /dev/null 2>&1)
head -3 $OUT/prof_eval/${RND}_kernel_stats.csv | cut -c1-200
echo "== tuples form (coherent rows)"
timeout 300 python bench.py --form tuples --cpu-baseline none 2>/dev/null | tail -1 > $OUT/bench_tuples.json; cut -c1-300 $OUT/bench_tuples.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats_tuples -o $RND -- python $R/bench.py --form tuples --steps 8 --warmup 2 --cpu-baseline none --also-relaxed 0 > /dev/null 2>&1)
head -3 $OUT/prof_stats_tuples/${RND}_kernel_stats.csv | cut -c1-250
echo "== f32 matrix pipe probe"
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/mfma_probe.hip 2>/dev/null && timeout 120 /tmp/mfma_probe 20000 > $OUT/mfma_probe.txt 2>&1; tail -3 $OUT/mfma_probe.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
echo "== done"
