#!/bin/bash
set +e
export TMPDIR=/tmp
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | grep -E "^E  |^tests/|passed|failed|Error" | cut -c1-260
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); rc=d.get('relaxed_coherence'); print('%-44s %8.2f Mw/s  frac %.3f  launch %.2f ms | relaxed %s' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms'], ('%.2fM/%.3f'%(rc['value']/1e6, rc['roofline_frac'])) if rc else None))
" "$1"; }
B="python bench.py --cpu-baseline none --also-relaxed 0 --tokens 50000000 --steps 10 --warmup 3"
timeout 600 $B 2>/dev/null | short "worker default (coherent, resident)"
timeout 600 $B --relaxed 1 --window-cache 1 2>/dev/null | short "worker resident relaxed"
timeout 600 $B --ids uniform 2>/dev/null | short "worker resident coherent uniform"
timeout 600 $B --dim 200 --vocab 60238 2>/dev/null | short "cfg1 shape coherent"
timeout 600 $B --form tuples 2>/dev/null | short "tuples coherent"
timeout 600 $B --form tuples --relaxed 1 2>/dev/null | short "tuples relaxed"
timeout 600 $B --relaxed 1 2>/dev/null | short "worker plain relaxed (auto)"
