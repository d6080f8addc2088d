#!/bin/bash
# round 5, session g: which part of the row-group path moves the two regimes that left their gates in session f --
# heldout_zipf12 at 256 workers and the planted corpus at the configs[2] shape with 8 workers -- under knob arms
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r05g
mkdir -p $OUT
for arm in "" "-refresh-rows -1" "-refresh-rows 4" "-refresh-rows 16" "-row-groups 0"; do
  echo "== arm [$arm]" | tee -a $OUT/arms.txt
  W2B_FIDELITY_EXTRA="$arm" timeout 300 python -m pytest tests/test_gpu_fidelity.py -q -m gpu -s -k "(held_out and zipf12) or (planted_matches and cfg2 and 8)" 2>&1 | grep -E "FIDELITY.*(auto|workers\))" | sed -e 's/losses.*deviation/deviation/' | tee -a $OUT/arms.txt
done
echo "== done"
