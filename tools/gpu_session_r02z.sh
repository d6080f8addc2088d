#!/bin/bash
# SQ counters of the evaluator's matrix kernel: where the wave cycles go (parked / issue-stalled / active), LDS, clock
set +e
export TMPDIR=/tmp
R=$PWD
CMD="python $R/bench.py --form eval --cpu-baseline none --steps 4 --warmup 1"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/prof_eval_sq -o sq -- $CMD > $R/gpurun_out/rocprof_eval_sq.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/prof_eval_sq2 -o sq -- $CMD > $R/gpurun_out/rocprof_eval_sq2.log 2>&1)
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/prof_eval_sq", "gpurun_out/prof_eval_sq2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, c in agg.items():
            if "eval_scores" in k:
                print(k, {cn: v / n[(k, cn)] for cn, v in c.items()})
PY
tail -3 gpurun_out/rocprof_eval_sq.log | cut -c1-300
