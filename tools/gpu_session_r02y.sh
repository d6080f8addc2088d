#!/bin/bash
set +e
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_eval.py -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -4 | cut -c1-300
for lib in "" _x256; do for dim in 200 808; do
  W2B_LIB=$PWD/word2bits_amd/libword2bits_hip$lib.so timeout 300 python bench.py --form eval --cpu-baseline none --dim $dim --steps 6 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', $dim, d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done; done
W2B_LIB=$PWD/word2bits_amd/libword2bits_hip_x128.so timeout 300 python tools/eval_ticks.py 200 2>&1 | tail -2
