#!/bin/bash
# bitlevel-2 quantizer in four instructions: parity + the bench legs
set +e
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_exact.py tests/test_gpu_worker.py -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -6 | cut -c1-300
timeout 600 python bench.py --cpu-baseline none 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('headline', d['value'], d['roofline']['frac'])
for k in ('with_loss_bookkeeping','bitlevel2','relaxed_coherence'):
    if k in d: print(k, d[k])
"
