#!/bin/bash
# same box, alternating: (b) round-2 loop with __syncthreads, (b2) the same with an LDS-only worker barrier,
# ("") early first-chunk loads + LDS-only barrier
set +e
export TMPDIR=/tmp
for rep in 1 2; do for lib in _b _b2 ""; do
  W2B_LIB=$PWD/word2bits_amd/libword2bits_hip$lib.so timeout 300 python bench.py --cpu-baseline none --also-relaxed 0 --also-legs 0 --steps 12 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 lib$lib', d['value'], d['roofline']['frac'])"
done; done
for lib in _b _b2 ""; do
  W2B_LIB=$PWD/word2bits_amd/libword2bits_hip$lib.so timeout 300 python bench.py --cpu-baseline none --also-relaxed 0 --also-legs 0 --steps 12 --vocab 3700000 --dim 1000 --negative 12 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 lib$lib', d['value'], d['roofline']['frac'])"
done
