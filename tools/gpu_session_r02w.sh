#!/bin/bash
# tuple kernel with private hot rows: parity tests, then the coherent tuples form with and without them
set +e
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_exact.py -m gpu -q --no-header -p no:cacheprovider --tb=short -s 2>&1 | grep -E "^E  |^tests/|passed|failed|Error|TUPLE HOT" | cut -c1-300
for hot in "0,0" "1,1" "2,2" "0,2" "2,0" "4,4"; do
  for per in 8 32; do
    echo "== W2B_TUPLE_HOT=$hot W2B_HOT_PERIOD=$per"
    W2B_TUPLE_HOT=$hot W2B_HOT_PERIOD=$per timeout 300 python bench.py --form tuples --cpu-baseline none --also-relaxed 0 --steps 6 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['ms_per_step'])"
  done
done
echo "== default plan"
timeout 300 python bench.py --form tuples --cpu-baseline none --steps 6 --warmup 2 2>&1 | tail -1
echo "== default plan, cfg5 shape"
timeout 300 python bench.py --form tuples --cpu-baseline none --also-relaxed 0 --vocab 3700000 --dim 1000 --negative 12 --window 5 --steps 6 --warmup 2 2>&1 | tail -1
