#!/bin/bash
# round 5, session d: EXPERIMENT -- what the hottest rows of each table cost the shared-row mode: rows 1..N of u read through
# the XCD's L2 (stale, fast), rows 1..N of v read and written through it (throughput only; not a training mode)
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r05d
mkdir -p $OUT
G="groups256:row_groups=1,workers=256,loss=1;groups512:row_groups=1,workers=512,loss=1"
for shape in "d200|--vocab 60238 --dim 200" "d800|"; do
  name="${shape%%|*}"; fl="${shape##*|}"
  for exp in "0 0" "32 0" "0 32" "32 32" "1 1"; do
    set -- $exp
    echo "== $name gx_u=$1 gx_v=$2" | tee -a $OUT/hotcost.txt
    W2B_EXP_GXU=$1 W2B_EXP_GXV=$2 timeout 200 python tests/experiments/arm_bench.py $fl --rounds 1 --arms "$G" 2>&1 | grep -A3 "== best" | tail -2 | tee -a $OUT/hotcost.txt
  done
  echo "== $name gx_u=32 gx_v=32, no adds" | tee -a $OUT/hotcost.txt
  W2B_EXP_GXU=32 W2B_EXP_GXV=32 timeout 200 python tests/experiments/arm_bench.py $fl --rounds 1 --arms "g256_noadd:row_groups=1,workers=256,loss=1,atomic_rank_u=-1" 2>&1 | grep -A3 "== best" | tail -1 | tee -a $OUT/hotcost.txt
done
echo "== done"
