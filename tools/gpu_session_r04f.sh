#!/bin/bash
# round 4, session f: the planted regimes (losses AND accuracy) under three arms of the lossless-row rule; exchange with the
# contributor mean for (nearly) every row.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r04f
mkdir -p $OUT
for arm in "" "-atomic-rank 0 -atomic-rank-u 2128" "-atomic-rank 0 -atomic-rank-u -1"; do
echo "== planted tests, extra flags: [$arm]"
W2B_FIDELITY_EXTRA="$arm" timeout 600 python -m pytest -q -m gpu tests/test_gpu_fidelity.py -k "planted_matches" -s 2>&1 | grep -E "FIDELITY|passed|failed" | cut -c1-330 | tee -a $OUT/planted_arms.txt
done
echo "== exchange: contributor mean from the first update on"
timeout 900 python tests/experiments/exchange_matrix.py --positions 256,64 --replicas 2,4 --tiers 8:16 --sat 1,8 --modes 2 > $OUT/exchange_matrix.txt 2>$OUT/exchange_matrix.err
grep XM $OUT/exchange_matrix.txt | cut -c1-170; tail -2 $OUT/exchange_matrix.err
echo "== done"
