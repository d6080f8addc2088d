#!/bin/bash
# round 4, session j: a HELD-OUT regime on a FULL device.  The per-XCD copies with the consensus rule are a balance measured
# on the benchmarked regime only; this runs the first held-out regime (V = 100 K, size 300, window 5, negative 5, -sample 0) on
# a 60 M-token stream, where -threads 0 fills the device, beside ONE run of the unmodified reference at 256 threads.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r04j
mkdir -p $OUT
python - <<'PY'
import sys, time
sys.path.insert(0, "tests")
from w2b_testlib import write_heldout_corpus
t = time.time(); write_heldout_corpus("/tmp/k5big.txt", "heldout_k5_big"); print("corpus written in %.0f s" % (time.time() - t))
PY
F="-bitlevel 1 -size 300 -window 5 -negative 5 -iter 1 -sample 0 -min-count 5 -binary 1"
( T0=$(date +%s); oracle/_ref/word2bits_stock -train /tmp/k5big.txt -output /dev/null -threads 256 $F > $OUT/ref_256.txt 2> /dev/null; echo "Elapsed $(( $(date +%s) - T0 )) s" > $OUT/ref_256.time; echo reference done ) &
REF=$!
for arm in "default|-threads 0" "no copies|-threads 0 -hot-rows 0" "768 workers|-threads 768" "512 workers|-threads 512" "256 workers|-threads 256" "default again|-threads 0"; do
  name="${arm%%|*}"; fl="${arm##*|}"
  T0=$(date +%s.%N)
  ./word2bits -train /tmp/k5big.txt -output /dev/null $F $fl > $OUT/run.txt 2> $OUT/run.err
  echo "K5BIG $name [$fl]: $(grep -o 'Hogwild workers (workgroups): [0-9]*' $OUT/run.txt) $(tr '\r' '\n' < $OUT/run.txt | grep 'Epoch Loss') ($(python -c "print('%.1f s' % ($(date +%s.%N) - $T0))"))" | tee -a $OUT/k5big.txt
done
wait $REF
tr '\r' '\n' < $OUT/ref_256.txt | grep -E "Vocab size|Words in train|Epoch Loss" | tee -a $OUT/k5big.txt
grep -E "Elapsed" $OUT/ref_256.time | tee -a $OUT/k5big.txt
rm -f /tmp/k5big.txt
echo "== done"
