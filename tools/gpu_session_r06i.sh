#!/bin/bash
# round 6, session i: planted corpus at the configs[2] shape, 64 workers: workers at once around the chosen 3/8 (24), twice each
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r06i
mkdir -p $OUT
timeout 900 python tests/experiments/planted_concurrency.py --cfg2-64 0,0,24,28,32,40,24,28,32,40 --out $OUT/planted_concurrency.json 2>&1 | grep PC | tee $OUT/planted_concurrency.txt
echo "== done"
