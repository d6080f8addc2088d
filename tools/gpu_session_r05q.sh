#!/bin/bash
# round 5, session q: between the reference's scale and a full device (257 .. 767 workers) the shared-row mode drifts on the
# benchmarked regime (+1.3 % at 440, +1.5 % at 512).  Round 4 had measured "4 copies of v, merged every word" at -0.1 % for 440
# workers and never adopted it (-threads 0 does not pick such counts).  Does it hold across the range?
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r05q
mkdir -p $OUT
python - <<'PY'
import sys
sys.path.insert(0, "tests")
from w2b_testlib import write_headline_corpus
write_headline_corpus("/tmp/headline.txt")
PY
F="-bitlevel 1 -size 800 -window 8 -negative 24 -iter 1 -sample 0 -min-count 5 -binary 1"
run() {
  ./word2bits -train /tmp/headline.txt -output /dev/null $F $1 > $OUT/run.txt 2> $OUT/run.err
  python - "$1" <<PY | tee -a $OUT/mid_range.txt
import re, sys
out = open("$OUT/run.txt").read().replace("\r", "\n")
L = [float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", out)]
m = re.findall(r"Words/thread/sec: ([\d.]+)k", out)
ref = -126013238.0
print("%-60s deviation %+.2f %%  (%sk words/thread/s)" % (sys.argv[1], 100 * (L[0] - ref) / abs(ref), m[-1] if m else "?"))
PY
}
for w in 320 440 512 640 767; do
  run "-threads $w"
  run "-threads $w -hot-rows-v 4 -hot-rows-u 0 -hot-period 1"
  run "-threads $w -hot-rows-v 8 -hot-rows-u 0 -hot-period 1"
done
rm -f /tmp/headline.txt
echo "== done"
