#!/bin/bash
# round 5, session i (as h, with W2B_EXP_TGT_LATE=1 arms): what separates the row-group kernel from the plain kernel on heldout_zipf12 at 256 workers (no copies):
# lossless target rows, worker counts
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r05i
mkdir -p $OUT
python - <<'PY'
import sys
sys.path.insert(0, "tests")
from w2b_testlib import write_heldout_corpus
write_heldout_corpus("/tmp/z12.txt", "heldout_zipf12")
PY
F="-bitlevel 2 -size 400 -window 8 -negative 24 -iter 2 -min-count 5 -binary 1"
REF1=-15222147; REF2=-12975108
run() {
  ./word2bits -train /tmp/z12.txt -output /dev/null $F $2 > $OUT/run.txt 2> $OUT/run.err
  python - "$1" "$2" <<PY | tee -a $OUT/z12.txt
import re, sys
out = open("$OUT/run.txt").read().replace("\r", "\n")
L = [float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", out)]
ref = [$REF1, $REF2]
print("%-34s %-44s deviation %s %%" % (sys.argv[1], sys.argv[2], ["%+.2f" % (100 * (a - b) / abs(b)) for a, b in zip(L, ref)]))
PY
}
for late in 0 1; do
  export W2B_EXP_TGT_LATE=$late
  for arm in "-threads 256 -refresh-rows -1" "-threads 256 -refresh-rows -1" "-threads 192 -refresh-rows -1" "-threads 128 -refresh-rows -1" "-threads 256 -refresh-rows 4" "-threads 256 -refresh-rows 0"; do
    run "groups late=$late" "-row-groups 1 $arm"
  done
done
run "plain" "-row-groups 0 -threads 256"
rm -f /tmp/z12.txt
echo "== done"
