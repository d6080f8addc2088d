#!/bin/bash
# round 4, session a: (1) atomic-add coherence / throughput probe and HBM-counter calibration, (2) GPU tests of what changed
# (loss bookkeeping at 13-row chunks, sc1 atomics, late round / fresh rows knobs), (3) throughput of the knob arms in one
# process, (4) reference bands of the HELD-OUT regimes on the box's host cores in the background while (5) the fidelity
# matrix of the benchmarked regime runs on the GPU, then the held-out regimes against the fresh bands.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r04a
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
echo "== (1) probes"
timeout 180 tools/atomic_probe 2>&1 | tee $OUT/atomic_probe.txt
(cd /tmp && timeout 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$OUT/cal_fetch -o cal -- $R/tools/row_probe calib > $R/$OUT/calib.txt 2>&1)
(cd /tmp && timeout 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$OUT/cal_write -o cal -- $R/tools/row_probe calib > $R/$OUT/calib_w.txt 2>&1)
grep -E "TB/s|CALIB" $OUT/calib.txt | head -20
python tools/pmc_calib.py $OUT/calib.txt $(find $OUT/cal_fetch -name "*counter_collection.csv" | head -1) $(find $OUT/cal_write -name "*counter_collection.csv" | head -1) $OUT/pmc_calibration.json
echo "== (2) tests"
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_parity.py tests/test_gpu_exact.py tests/test_gpu_worker.py tests/test_gpu_integration.py tests/test_gpu_cli.py 2>&1 | tail -15 | tee $OUT/pytest_quick.txt
echo "== (3) arms: throughput"
timeout 900 python tests/experiments/arm_bench.py --rounds 2 --out $OUT/arm_bench.json --arms "default:;loss:loss=1;late:hot_late=1;late_loss:hot_late=1,loss=1;fresh128:fresh_rank_u=128;fresh2000:fresh_rank_u=2000;atomu300:atomic_rank_u=300,atomic_rank=0;atomu3000:atomic_rank_u=3000,atomic_rank=0;atomv300:atomic_rank=300,atomic_rank_u=-1;atomv3000:atomic_rank=3000,atomic_rank_u=-1;p8:hot_period=8;resident:window_cache=1" 2>&1 | tee $OUT/arm_bench.txt
echo "== (4) held-out reference bands on the host (background)"
(timeout 1500 python tests/golden/make_fidelity_bands.py --out $OUT/bands_heldout.json --jobs heldout_k5,heldout_zipf12 --heldout 64x2,256x2 > $OUT/bands_heldout.log 2>&1; echo bands done) &
BANDS=$!
echo "== (5) fidelity matrix, benchmarked regime"
ARMS="default:;late:-hot-late 1;fresh128:-fresh-rank-u 128;fresh2000:-fresh-rank-u 2000;fresh2000+late:-fresh-rank-u 2000 -hot-late 1;atomu3000:-atomic-rank-u 3000 -atomic-rank 0;atomv1000:-atomic-rank 1000 -atomic-rank-u -1;atom_uv:-atomic-rank 1000 -atomic-rank-u 3000;atom_uv+late:-atomic-rank 1000 -atomic-rank-u 3000 -hot-late 1;fresh+atomv+late:-fresh-rank-u 3000 -atomic-rank 1000 -atomic-rank-u -1 -hot-late 1;hot40:-hot-rows 40;hot40+fresh+late:-hot-rows 40 -fresh-rank-u 2000 -hot-late 1"
timeout 1500 python tests/experiments/fidelity_matrix.py --jobs headline --threads 0,256,64 --kernel plain --out $OUT/fidelity.jsonl --arms "$ARMS" 2>&1 | tee $OUT/fidelity_headline.txt | cut -c1-220
wait $BANDS
tail -3 $OUT/bands_heldout.log | cut -c1-300
echo "== (6) fidelity matrix, held-out regimes"
ARMS2="default:;fresh2000+late:-fresh-rank-u 2000 -hot-late 1;atom_uv+late:-atomic-rank 1000 -atomic-rank-u 3000 -hot-late 1"
timeout 900 python tests/experiments/fidelity_matrix.py --jobs heldout_k5,heldout_zipf12 --threads 0,256,64 --kernel both --bands $OUT/bands_heldout.json --out $OUT/fidelity.jsonl --arms "$ARMS2" 2>&1 | tee $OUT/fidelity_heldout.txt | cut -c1-220
rm -rf $OUT/cal_fetch $OUT/cal_write
echo "== done"
