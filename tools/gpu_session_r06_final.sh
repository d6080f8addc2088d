#!/bin/bash
# round 6, final session: (1) the driver's bench command; (2) rocprofv3 kernel stats and the PMC passes of the same command;
# (3) kernel stats of the other shapes; (4) counters for the two short-row legs of the row-group kernel (round-5 review: "no traffic
# evidence for this kernel at all"); (5) the evaluator; (6) smoke().
set +e
RND=${RND:-r06}
OUT=gpurun_out/${RND}_final
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== (1) bench (the driver's command)"
T0=$(date +%s)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.log 2>$OUT/bench_default.err
echo "bench wall: $(( $(date +%s) - T0 )) s"
tail -1 $OUT/bench_default.log | cut -c1-1200
K=$(tail -1 $OUT/bench_default.log | python -c "import json,sys; print(json.loads(sys.stdin.read())['roofline']['kernel'])")
CMD="python $R/bench.py --steps 8 --warmup 2 --cpu-baseline none --cpu-cfg0 0 --also-relaxed 0 --also-legs 0 --also-shapes 0"
echo "== (2) rocprofv3 of: $CMD   (kernel $K)"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats -o $RND -- $CMD > $R/$OUT/rocprof_stats.log 2>&1)
grep "^{" $OUT/rocprof_stats.log | tail -1 > $OUT/bench_profiled.json
head -4 $OUT/prof_stats/${RND}_kernel_stats.csv | cut -c1-250
pmc() {   # name, kernel substring, extra bench flags
  n=$1; ks=$2; shift; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 900 rocprofv3 --pmc $c --output-format csv -d $R/$OUT/prof_${n}_$c -o $RND -- $CMD "$@" > $R/$OUT/rocprof_${n}_$c.log 2>&1)
  done
  (cd /tmp && timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$OUT/prof_${n}_l2 -o $RND -- $CMD "$@" > $R/$OUT/rocprof_${n}_l2.log 2>&1)
  grep "^{" $OUT/rocprof_${n}_FETCH_SIZE.log | tail -1 > $OUT/bench_pmc_$n.json
  python tools/pmc_summary.py $OUT/prof_${n}_FETCH_SIZE/${RND}_counter_collection.csv $OUT/prof_${n}_WRITE_SIZE/${RND}_counter_collection.csv $OUT/prof_${n}_l2/${RND}_counter_collection.csv $OUT/pmc_$n.json $ks $OUT/bench_pmc_$n.json | cut -c1-700
}
pmc worker $K
echo "== (3) kernel stats of the other shapes"
S="--steps 8 --warmup 2 --tokens 30000000 --cpu-baseline none --cpu-cfg0 0 --also-relaxed 0 --also-legs 0 --also-shapes 0"
prof() { name=$1; shift; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats_$name -o $RND -- python $R/bench.py $S "$@" > $R/$OUT/rocprof_stats_$name.log 2>&1); tail -1 $OUT/rocprof_stats_$name.log | cut -c1-160; head -3 $OUT/prof_stats_$name/${RND}_kernel_stats.csv | tail -2 | cut -c1-200; }
prof d200 --vocab 60238 --dim 200
prof d400b2 --vocab 60238 --dim 400 --bitlevel 2
prof cfg5 --vocab 3700000 --dim 1000 --negative 12 --tokens 60000000
prof cfg5_resident --vocab 3700000 --dim 1000 --negative 12 --tokens 60000000 --window-cache 1
prof tuples --form tuples
echo "== (4) counters: the row-group kernel's short-row legs"
pmc d200 k_train_groups --tokens 30000000 --vocab 60238 --dim 200
pmc d400b2 k_train_groups --tokens 30000000 --vocab 60238 --dim 400 --bitlevel 2
echo "== (5) evaluator"
timeout 300 python bench.py --form eval --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/eval_bench.json; cut -c1-300 $OUT/eval_bench.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_eval -o $RND -- python $R/bench.py --form eval --steps 5 --warmup 1 --eval-cpu-questions 0 > /dev/null 2>&1)
head -3 $OUT/prof_eval/${RND}_kernel_stats.csv | cut -c1-200
echo "== (6) smoke"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/kernel_resources.sh w2b_kernels_workers.hip "k_train_workers<1, 4, true, 256, 0" > $OUT/kernel_resources.txt 2>/dev/null; bash tools/kernel_resources.sh w2b_kernels_groups.hip "k_train_groups<1, true\|k_train_groups<2, true" >> $OUT/kernel_resources.txt 2>/dev/null; cat $OUT/kernel_resources.txt | cut -c1-220
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete; find $OUT -name "*agent_info.csv" -delete
echo "== done"
