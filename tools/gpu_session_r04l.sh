#!/bin/bash
# round 4, session l: the word-count / alpha period scaled beyond 256 workers (total lag of a 256-thread run): fidelity at the
# worker counts that change (> 256), every regime.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r04l
mkdir -p $OUT
timeout 900 python tests/experiments/fidelity_matrix.py --jobs text8size --threads 850,340 --kernel auto --out $OUT/fidelity.jsonl --arms "default:" 2>&1 | tee $OUT/fidelity.txt | cut -c1-200
timeout 900 python tests/experiments/fidelity_matrix.py --jobs headline --threads 1024,768,440 --kernel auto --out $OUT/fidelity.jsonl --arms "default:" 2>&1 | tee -a $OUT/fidelity.txt | cut -c1-200
timeout 900 python tests/experiments/fidelity_matrix.py --jobs heldout_zipf12 --threads 296 --kernel auto --out $OUT/fidelity.jsonl --arms "default:" 2>&1 | tee -a $OUT/fidelity.txt | cut -c1-200
timeout 900 python tests/experiments/fidelity_matrix.py --jobs heldout_k5 --threads 425 --kernel auto --out $OUT/fidelity.jsonl --arms "default:" 2>&1 | tee -a $OUT/fidelity.txt | cut -c1-200
timeout 600 python -m pytest -x -q -m gpu tests/test_gpu_fidelity.py -k "full_device" -s 2>&1 | grep -E "FIDELITY|passed|failed" | cut -c1-250 | tee -a $OUT/fidelity.txt
echo "== done"
