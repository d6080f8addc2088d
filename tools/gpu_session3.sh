#!/bin/bash
set +e
OUT=gpurun_out/s3
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider 2>&1 | grep -vE "^\s*$" | tail -25 | cut -c1-250 | tee $OUT/pytest.log
echo "== accuracy bitlevel 1"; timeout 1200 python tools/accuracy_experiment.py --bitlevel 1 2>&1 | tee $OUT/acc_b1.log
echo "== accuracy bitlevel 2"; timeout 900 python tools/accuracy_experiment.py --bitlevel 2 --cpu-threads 8 --gpu-threads 8,1024 2>&1 | tee $OUT/acc_b2.log
echo "== bench"; timeout 900 python bench.py > $OUT/bench_tuples.log 2>$OUT/bench_tuples.err; tail -1 $OUT/bench_tuples.log | cut -c1-1500
timeout 600 python bench.py --form worker --cpu-baseline none > $OUT/bench_worker.log 2>/dev/null; tail -1 $OUT/bench_worker.log | cut -c1-400
W2B_LIB=$PWD/word2bits_amd/libword2bits_hip_sc1.so timeout 600 python bench.py --cpu-baseline none --ids uniform > $OUT/bench_sc1_uniform.log 2>/dev/null; tail -1 $OUT/bench_sc1_uniform.log | cut -c1-400
timeout 600 python bench.py --cpu-baseline none --ids uniform > $OUT/bench_uniform.log 2>/dev/null; tail -1 $OUT/bench_uniform.log | cut -c1-400
echo "== rocprof stats"
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_stats -o r01 -- python $R/bench.py --steps 8 --warmup 2 --cpu-baseline none > $R/$OUT/rocprof_stats.log 2>&1)
find $OUT/prof_stats -name "*kernel_stats*" | head -3; for f in $(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1); do head -6 $f | cut -c1-300; done
echo "== rocprof pmc fetch"
(cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/prof_fetch -o r01 -- python $R/bench.py --steps 4 --warmup 1 --cpu-baseline none > $R/$OUT/rocprof_fetch.log 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/prof_write -o r01 -- python $R/bench.py --steps 4 --warmup 1 --cpu-baseline none > $R/$OUT/rocprof_write.log 2>&1)
(cd /tmp && timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/$OUT/prof_l2 -o r01 -- python $R/bench.py --steps 4 --warmup 1 --cpu-baseline none > $R/$OUT/rocprof_l2.log 2>&1)
for d in prof_fetch prof_write prof_l2; do f=$(find $OUT/$d -name "*counter_collection.csv" | head -1); echo $f; python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
if f:
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r.get("Kernel_Name", "")[:60], r.get("Counter_Name"))
        agg[k][0] += 1; agg[k][1] += float(r.get("Counter_Value", 0))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:6]:
        print(k, "dispatches", v[0], "sum %.4g" % v[1], "per-dispatch %.4g" % (v[1] / v[0]))
PY
done
# keep merge-back small: drop the big per-dispatch traces, keep stats
find $OUT -name "*kernel_trace.csv" -size +2M -delete; find $OUT -name "*.db" -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
echo "== done"
