#!/bin/bash
# hot-row merges staggered across workers vs all workers at the same steps, same box, alternating
set +e
export TMPDIR=/tmp
for rep in 1 2; do for lib in "" _ns; do
  W2B_LIB=$PWD/word2bits_amd/libword2bits_hip$lib.so timeout 300 python bench.py --cpu-baseline none --also-relaxed 0 --also-legs 0 --steps 12 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib$lib', d['value'], d['roofline']['frac'])"
done; done
