#!/bin/bash
# round 5, session b: where the shared-row mode's ~13 M words/s cap comes from -- row-group kernel, uniform ids (no hot rows),
# Zipf ids with and without the lossless adds, a Zipf stream without its hottest words
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r05b
mkdir -p $OUT
G="groups256:row_groups=1,workers=256,loss=1;groups512:row_groups=1,workers=512,loss=1;plain256:row_groups=0,workers=256,loss=1"
N="g256_noadd:row_groups=1,workers=256,loss=1,atomic_rank_u=-1;g512_noadd:row_groups=1,workers=512,loss=1,atomic_rank_u=-1;p256_noadd:row_groups=0,workers=256,loss=1,atomic_rank_u=-1"
for shape in "d200|--vocab 60238 --dim 200" "d800|"; do
  name="${shape%%|*}"; fl="${shape##*|}"
  echo "== $name uniform ids" | tee -a $OUT/cap.txt
  timeout 200 python tests/experiments/arm_bench.py $fl --ids uniform --rounds 1 --arms "$G" 2>&1 | grep -A9 "== best" | tee -a $OUT/cap.txt
  echo "== $name zipf, lossless u / plain stores" | tee -a $OUT/cap.txt
  timeout 200 python tests/experiments/arm_bench.py $fl --rounds 1 --arms "$G;$N" 2>&1 | grep -A9 "== best" | tee -a $OUT/cap.txt
  echo "== $name zipf without its 16 / 128 hottest words" | tee -a $OUT/cap.txt
  timeout 200 python tests/experiments/arm_bench.py $fl --zipf-shift 16 --rounds 1 --arms "$G" 2>&1 | grep -A9 "== best" | tee -a $OUT/cap.txt
  timeout 200 python tests/experiments/arm_bench.py $fl --zipf-shift 128 --rounds 1 --arms "$G" 2>&1 | grep -A9 "== best" | tee -a $OUT/cap.txt
done
echo "== done"
