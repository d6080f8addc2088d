#!/bin/bash
# hot rows: how many, merged how often?  speed on the default bench x first-epoch loss at text8 size
set +e
export TMPDIR=/tmp
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-40s %8.2f Mw/s  frac %.3f' % (sys.argv[1], d['value']/1e6, d['roofline']['frac']))
" "$1"; }
B="timeout 300 python bench.py --cpu-baseline none --also-relaxed 0 --also-legs 0 --tokens 50000000 --steps 8 --warmup 2"
for nh in 0 2 4 6; do for p in 4 8 16 32; do
  [ $nh = 0 ] && [ $p != 4 ] && continue
  W2B_HOT_ROWS=$nh W2B_HOT_PERIOD=$p $B 2>/dev/null | short "bench: hot $nh period $p"
done; done
python - <<'PY'
import sys; sys.path.insert(0, "tests")
from w2b_testlib import write_zipf_text_corpus
write_zipf_text_corpus("/tmp/t8.txt")
PY
run() { echo -n "text8size $1: "; shift; env "$@" ./word2bits -train /tmp/t8.txt -output /tmp/o.bin -bitlevel 1 -size 200 -window 8 -negative 24 -iter 2 -min-count 5 -binary 1 -threads 0 2>&1 | tr '\r' '\n' | grep -E "Epoch Loss|Words/thread" | tail -3 | tr '\n' ' ' | sed 's/Alpha[^W]*//g'; echo; }
for nh in 2 4 6; do for p in 4 8 16; do run "hot $nh period $p" W2B_HOT_ROWS=$nh W2B_HOT_PERIOD=$p; done; done
