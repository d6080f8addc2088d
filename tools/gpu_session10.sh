#!/bin/bash
set +e
OUT=gpurun_out/s10
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | grep -E "^E  |^tests/|passed|failed|Error" | cut -c1-260 | tee $OUT/pytest.log
echo "== bench default"; timeout 1200 python bench.py > $OUT/bench_default.log 2>$OUT/bench_default.err; tail -1 $OUT/bench_default.log | cut -c1-3000
echo "== bench tuples"; timeout 900 python bench.py --form tuples --cpu-baseline none > $OUT/bench_tuples.log 2>/dev/null; tail -1 $OUT/bench_tuples.log | cut -c1-1700
echo "== bench worker uniform ids"; timeout 900 python bench.py --ids uniform --cpu-baseline none > $OUT/bench_uniform.log 2>/dev/null; tail -1 $OUT/bench_uniform.log | cut -c1-1700
echo "== 1-rank torchrun"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 6 --warmup 2 --cpu-baseline none --also-relaxed 0 --tokens 20000000 2>&1 | tail -1 | cut -c1-300
for form in worker tuples; do
  K=k_train_workers2; [ $form = tuples ] && K=k_train_tuples
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats_$form -o r01 -- python $R/bench.py --form $form --steps 8 --warmup 2 --cpu-baseline none --also-relaxed 0 > $R/$OUT/rocprof_stats_$form.log 2>&1)
  (cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/prof_fetch_$form -o r01 -- python $R/bench.py --form $form --steps 4 --warmup 1 --cpu-baseline none --also-relaxed 0 > $R/$OUT/rocprof_fetch_$form.log 2>&1)
  (cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/prof_write_$form -o r01 -- python $R/bench.py --form $form --steps 4 --warmup 1 --cpu-baseline none --also-relaxed 0 > $R/$OUT/rocprof_write_$form.log 2>&1)
  (cd /tmp && timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$OUT/prof_l2_$form -o r01 -- python $R/bench.py --form $form --steps 4 --warmup 1 --cpu-baseline none --also-relaxed 0 > $R/$OUT/rocprof_l2_$form.log 2>&1)
  python tools/pmc_summary.py $OUT/prof_fetch_$form/r01_counter_collection.csv $OUT/prof_write_$form/r01_counter_collection.csv $OUT/prof_l2_$form/r01_counter_collection.csv $OUT/pmc_$form.json $K | cut -c1-500
  head -3 $OUT/prof_stats_$form/r01_kernel_stats.csv | cut -c1-250
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
echo "== accuracy / loss fidelity table (default = coherent, automatic kernel)"
timeout 1200 python tools/accuracy_experiment.py --bitlevel 1 --cpu-threads 1,8,64 --gpu-threads 1,8,64,512 --variants coherent,relaxed 2>&1 | cut -c1-330 | tee $OUT/acc_b1.log
timeout 900 python tools/accuracy_experiment.py --bitlevel 2 --cpu-threads 8 --gpu-threads 8,512 --variants coherent 2>&1 | cut -c1-330 | tee $OUT/acc_b2.log
echo "== done"
