#!/bin/bash
set +e
export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short -x 2>&1 | grep -E "^E  |^tests/|passed|failed|Error" | cut -c1-260
B="python bench.py --cpu-baseline none --also-relaxed 0 --tokens 30000000 --steps 8 --warmup 2"
W2B_LIB=$PWD/word2bits_amd/libword2bits_hip_dbg.so W2B_DEBUG=1 timeout 600 $B --form worker --workers 512 --positions 1024 2>&1 | grep -E "w2b debug" | cut -c1-400
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-40s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
for wc in 1 0; do for rl in 0 1; do
  timeout 600 $B --form worker --window-cache $wc --relaxed $rl 2>/dev/null | short "worker zipf wc=$wc relaxed=$rl"
done; done
timeout 600 $B --form worker --workers 512 --positions 2048 2>/dev/null | short "worker zipf wc=1 workers=512"
timeout 600 $B 2>/dev/null | short "tuples zipf coherent"
timeout 600 $B --relaxed 1 2>/dev/null | short "tuples zipf relaxed"
