#!/bin/bash
set +e
export TMPDIR=/tmp
python - <<'PY'
import sys; sys.path.insert(0, "tests")
from w2b_testlib import write_zipf_text_corpus
write_zipf_text_corpus("/tmp/t8.txt")
PY
run() { echo -n "text8size $1: "; shift; env "$@" ./word2bits -train /tmp/t8.txt -output /tmp/o.bin -bitlevel 1 -size 200 -window 8 -negative 24 -iter 2 -min-count 5 -binary 1 $TH 2>&1 | tr '\r' '\n' | grep -E "Epoch Loss" | tr '\n' ' '; echo; }
TH="-threads 0"
for nh in 4 6; do for p in 8 16 32; do run "850 workers hot $nh period $p" W2B_HOT_ROWS=$nh W2B_HOT_PERIOD=$p; done; done
TH="-threads 128"
for p in 8 16 32; do run "128 workers hot 6 period $p" W2B_HOT_ROWS=6 W2B_HOT_PERIOD=$p; done
run "128 workers hot 0" W2B_HOT_ROWS=0
TH="-threads 128 -window-cache 0"; run "128 workers plain" X=1
