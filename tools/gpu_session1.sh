#!/bin/bash
# First GPU session: environment probe, coherence probe, parity tests, smoke, first bench + rocprof.
# Everything is logged under gpurun_out/s1/ ; failures of one stage do not stop the next.
set +e
OUT=gpurun_out/s1
mkdir -p $OUT
export TMPDIR=/tmp
{
  echo "== env"; nproc; lscpu | grep -E "Model name|Socket|Thread|Core" ; free -g | head -2
  rocm-smi --showproductname 2>/dev/null | head -20
  ls /root/reference 2>&1 | head -2
  python -c "import torch; print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count)"
} > $OUT/env.log 2>&1
echo "== coherence probe"; timeout 120 ./tools/coherence_probe > $OUT/coherence.log 2>&1; cat $OUT/coherence.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -5 $OUT/smoke.log
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q -x --no-header -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -30 $OUT/pytest_gpu.log
echo "== pytest gpu (continue past first failure)"; timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $OUT/pytest_gpu_all.log 2>&1; tail -15 $OUT/pytest_gpu_all.log
echo "== bench tuples"; timeout 900 python bench.py --steps 12 --warmup 3 --cpu-baseline none > $OUT/bench_tuples.log 2>&1; tail -3 $OUT/bench_tuples.log
echo "== bench worker"; timeout 900 python bench.py --form worker --steps 12 --warmup 3 --cpu-baseline none > $OUT/bench_worker.log 2>&1; tail -3 $OUT/bench_worker.log
echo "== bench tuples uniform"; timeout 900 python bench.py --ids uniform --steps 12 --warmup 3 --cpu-baseline none > $OUT/bench_tuples_uniform.log 2>&1; tail -3 $OUT/bench_tuples_uniform.log
echo "== rocprof"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cpu-baseline none > $GRAFT_REPO_ROOT/$OUT/rocprof.log 2>&1); tail -3 $OUT/rocprof.log
find $OUT/prof -name "*stats*" | head; for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -8 $f; done
# keep the merge-back small
find $OUT/prof -name "*.csv" -size +4M -delete
echo "== done"
