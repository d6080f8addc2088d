#!/bin/bash
# round 5, session n: the full-device balance (per-XCD copies, consensus merges) on the LITERAL configs[1] stream (100 M tokens)
# beside its 22 M-token proxy: merge period 32 (default) / 16 / 8, consensus weight 1/8 (default) / 1/4
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r05n
mkdir -p $OUT
python - <<'PY'
import sys
sys.path.insert(0, "tests")
from w2b_testlib import write_headline_corpus
write_headline_corpus("/tmp/cfg1_100m.txt", n_zipf=98_000_000); write_headline_corpus("/tmp/headline.txt")
PY
F="-bitlevel 1 -size 800 -window 8 -negative 24 -iter 1 -sample 0 -min-count 5 -binary 1 -threads 1024"
run() {   # file, ref loss, flags
  ./word2bits -train $1 -output /dev/null $F $3 > $OUT/run.txt 2> $OUT/run.err
  python - "$1" "$2" "$3" <<PY | tee -a $OUT/balance.txt
import re, sys
out = open("$OUT/run.txt").read().replace("\r", "\n")
L = [float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", out)]
m = re.findall(r"Words/thread/sec: ([\d.]+)k", out)
ref = float(sys.argv[2])
print("%-22s %-34s loss %.0f deviation %+.2f %%  (%sk words/thread/s)" % (sys.argv[1], sys.argv[3], L[0], 100 * (L[0] - ref) / abs(ref), m[-1] if m else "?"))
PY
}
for arm in "" "-hot-period 16" "-hot-period 8" "-hot-weight 250" "-hot-period 16 -hot-weight 250" "-hot-period 64"; do
  run /tmp/cfg1_100m.txt -543450078.458814 "$arm"
  run /tmp/headline.txt -126013238 "$arm"
done
rm -f /tmp/cfg1_100m.txt /tmp/headline.txt
echo "== done"
