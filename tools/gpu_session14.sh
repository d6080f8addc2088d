#!/bin/bash
set +e
OUT=gpurun_out/s14
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --form worker --steps 3 --warmup 1 --cpu-baseline none --also-relaxed 0 --tokens 30000000"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $R/$OUT/p1 -o r -- $B > $R/$OUT/p1.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$OUT/p2 -o r -- $B > $R/$OUT/p2.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC SQ_WAVE32_INSTS --output-format csv -d $R/$OUT/p3 -o r -- $B > $R/$OUT/p3.log 2>&1)
for p in p1 p2 p3; do python - $OUT/$p/r_counter_collection.csv <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "k_train_workers2" in r.get("Kernel_Name", ""):
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, v in agg.items(): print("%-24s per-dispatch %.4g" % (k, v[1] / v[0]))
except Exception as e: print("ERR", e)
PY
tail -2 $OUT/$p.log | cut -c1-200
done
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
