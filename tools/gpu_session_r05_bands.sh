#!/bin/bash
# round 5, the long session: the reference bands the round-4 review asked for, on the GPU box's HOST (256 hardware threads) --
# BASELINE configs[1] literally (100 M tokens, one 256-thread run, ~13 min) and a second run of heldout_k5_big (~5.5 min) --
# while the GPU runs (a) ./word2bits on the same 100 M-token file at the bench's own 1024 workers, (b) the benchmarked regime
# at 440 / 768 workers, (c) the 8-replica configs[3] experiment, (d) the whole -m gpu suite.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r05m
mkdir -p $OUT /tmp/w2b_bands
( python tests/golden/make_fidelity_bands.py --out $OUT/bands_cfg1.json --jobs cfg1_100m --cfg1 256x1 --tmp /tmp/w2b_bands_a > $OUT/bands_cfg1.log 2>&1;
  python tests/golden/make_fidelity_bands.py --out $OUT/bands_k5big.json --jobs heldout_k5_big --heldout-big 256x1 --tmp /tmp/w2b_bands_b > $OUT/bands_k5big.log 2>&1;
  echo "reference bands done" ) &
REF=$!
python - <<'PY'
import sys, time
sys.path.insert(0, "tests")
from w2b_testlib import write_headline_corpus
t = time.time(); write_headline_corpus("/tmp/cfg1_100m.txt", n_zipf=98_000_000); print("cfg1 corpus written in %.0f s" % (time.time() - t))
t = time.time(); write_headline_corpus("/tmp/headline.txt"); print("headline corpus written in %.0f s" % (time.time() - t))
PY
F="-bitlevel 1 -size 800 -window 8 -negative 24 -iter 1 -sample 0 -min-count 5 -binary 1"
run() {   # name, file, flags
  T0=$(date +%s.%N)
  ./word2bits -train $2 -output /dev/null $F $3 > $OUT/run.txt 2> $OUT/run.err
  echo "RUN $1 [$3]: $(grep -o 'Hogwild workers (workgroups): [0-9]*' $OUT/run.txt) $(tr '\r' '\n' < $OUT/run.txt | grep 'Epoch Loss') ($(python -c "print('%.1f s' % ($(date +%s.%N) - $T0))"))" | tee -a $OUT/gpu_runs.txt
}
for arm in "-threads 1024" "-threads 0" "-threads 1024" "-threads 0 -hot-rows 0" "-threads 256"; do run cfg1_100m /tmp/cfg1_100m.txt "$arm"; done
for arm in "-threads 440" "-threads 768" "-threads 440" "-threads 768" "-threads 1024" "-threads 256" "-threads 512"; do run headline22m /tmp/headline.txt "$arm"; done
timeout 900 python tests/experiments/replicas8_cfg3.py /tmp/cfg1_100m.txt --out $OUT/replicas8.json > $OUT/replicas8.txt 2>&1; grep R8 $OUT/replicas8.txt
rm -f /tmp/cfg1_100m.txt /tmp/headline.txt
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
wait $REF
cat $OUT/bands_cfg1.log $OUT/bands_k5big.log | tail -6
echo "== done"
