#!/bin/bash
set +e
OUT=gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | grep -E "^E  |^tests/|passed|failed|Error" | cut -c1-260
echo "== bench default"; timeout 1200 python bench.py > $OUT/bench_default.log 2>$OUT/bench_default.err; tail -1 $OUT/bench_default.log | cut -c1-400
form=worker; K=k_train_workers2
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats_$form -o r01 -- python $R/bench.py --form $form --steps 8 --warmup 2 --cpu-baseline none --also-relaxed 0 > $R/$OUT/rocprof_stats_$form.log 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/prof_fetch_$form -o r01 -- python $R/bench.py --form $form --steps 4 --warmup 1 --cpu-baseline none --also-relaxed 0 > $R/$OUT/rocprof_fetch_$form.log 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/prof_write_$form -o r01 -- python $R/bench.py --form $form --steps 4 --warmup 1 --cpu-baseline none --also-relaxed 0 > $R/$OUT/rocprof_write_$form.log 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$OUT/prof_l2_$form -o r01 -- python $R/bench.py --form $form --steps 4 --warmup 1 --cpu-baseline none --also-relaxed 0 > $R/$OUT/rocprof_l2_$form.log 2>&1)
python tools/pmc_summary.py $OUT/prof_fetch_$form/r01_counter_collection.csv $OUT/prof_write_$form/r01_counter_collection.csv $OUT/prof_l2_$form/r01_counter_collection.csv $OUT/pmc_$form.json $K | cut -c1-500
head -3 $OUT/prof_stats_$form/r01_kernel_stats.csv | cut -c1-250
# relaxed plain worker kernel stats too (the side measurement of the bench line)
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats_relaxed -o r01 -- python $R/bench.py --relaxed 1 --steps 8 --warmup 2 --cpu-baseline none --also-relaxed 0 > $R/$OUT/rocprof_stats_relaxed.log 2>&1)
head -2 $OUT/prof_stats_relaxed/r01_kernel_stats.csv | cut -c1-250
# the evaluator's scan (include/word2bits_eval.h)
timeout 300 python tools/eval_bench.py 2>/dev/null | tail -1 > $OUT/eval_bench.json; cut -c1-300 $OUT/eval_bench.json
W2B_EVAL_KERNEL=0 timeout 300 python tools/eval_bench.py --eval-cpu-questions 0 2>/dev/null | tail -1 > $OUT/eval_bench_valu.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_eval -o r01 -- python $R/tools/eval_bench.py --eval-cpu-questions 0 > /dev/null 2>&1)
head -3 $OUT/prof_eval/r01_kernel_stats.csv | cut -c1-200
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
echo "== done"
