#!/bin/bash
set +e
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_eval.py -m gpu -q --no-header -p no:cacheprovider --tb=short -x 2>&1 | tail -5 | cut -c1-300
for dim in 104 200 400 808; do
  timeout 300 python bench.py --form eval --cpu-baseline none --dim $dim --steps 6 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($dim, d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['answers_differ_between_modes'])"
done
