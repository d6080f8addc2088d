#!/bin/bash
# round 4, closing session on the final tree: the whole -m gpu suite, smoke, the driver's bench command (timed).
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r04m
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.log 2>$OUT/bench_default.err ) 2> $OUT/time.txt
grep real $OUT/time.txt; tail -1 $OUT/bench_default.log | cut -c1-400
echo "== done"
