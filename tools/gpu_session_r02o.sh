#!/bin/bash
# fidelity of the sentence-resident kernel at text8 size (17 M tokens, 70 K words, D=200, -threads 0): where does the first-epoch loss go?
set +e
export TMPDIR=/tmp
python - <<'PY'
import sys; sys.path.insert(0, "tests")
from w2b_testlib import write_zipf_text_corpus
write_zipf_text_corpus("/tmp/t8.txt")
PY
run() { echo "== $1"; shift; env "$@" ./word2bits -train /tmp/t8.txt -output /tmp/o.bin -bitlevel 1 -size 200 -window 8 -negative 24 -iter 3 -min-count 5 -binary 1 $EXTRA 2>&1 | tr '\r' '\n' | grep -E "Hogwild|Epoch Loss" | tr '\n' ' '; echo; }
EXTRA="-threads 0 -window-cache 0" run "plain, threads 0" X=1
EXTRA="-threads 0" run "resident, hot auto, period 32" X=1
EXTRA="-threads 0" run "resident, hot 0" W2B_HOT_ROWS=0
EXTRA="-threads 0" run "resident, hot auto, period 4" W2B_HOT_PERIOD=4
EXTRA="-threads 0" run "resident, hot auto, period 1" W2B_HOT_PERIOD=1
EXTRA="-threads 64" run "resident, 64 workers" X=1
EXTRA="-threads 64 -window-cache 0" run "plain, 64 workers" X=1
EXTRA="-threads 256" run "resident, 256 workers" X=1
EXTRA="-threads 256 -window-cache 0" run "plain, 256 workers" X=1
