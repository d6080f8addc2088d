#!/bin/bash
set +e
OUT=gpurun_out/s11
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest new"; timeout 900 python -m pytest tests/test_gpu_parity.py::test_suggested_threads_fills_the_device tests/test_gpu_cli.py -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | grep -E "^E  |^tests/|passed|failed|Error" | cut -c1-260
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); rc=d.get('relaxed_coherence'); print('%-44s %8.2f Mw/s  frac %.3f  launch %.2f ms  steps %d | relaxed %s' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['steps'], ('%.2fM/%.3f'%(rc['value']/1e6, rc['roofline_frac'])) if rc else None))
" "$1"; }
echo "== long run (epoch restarts inside the timed region)"
timeout 900 python bench.py --gpus 1 --steps 150 --warmup 5 --cpu-baseline none 2>/dev/null | short "default, 150 steps"
echo "== cfg5 shape: V=3.7M D=1000 K=12 (bytes/word 176088)"
timeout 1200 python bench.py --vocab 3700000 --dim 1000 --negative 12 --bitlevel 1 --cpu-baseline none 2>$OUT/cfg5_b1.err | short "cfg5 bitlevel 1 worker"
timeout 1200 python bench.py --vocab 3700000 --dim 1000 --negative 12 --bitlevel 0 --cpu-baseline none 2>$OUT/cfg5_b0.err | short "cfg5 bitlevel 0 worker"
timeout 1200 python bench.py --vocab 3700000 --dim 1000 --negative 12 --bitlevel 1 --form tuples --cpu-baseline none 2>>$OUT/cfg5_b1.err | short "cfg5 bitlevel 1 tuples"
tail -2 $OUT/cfg5_b1.err | cut -c1-300
echo "== cfg3 shape: bitlevel 2 D=400 on the planted corpus, accuracy vs reference"
timeout 900 python tools/accuracy_experiment.py --bitlevel 2 --size 400 --cpu-threads 8 --gpu-threads 8,64 --variants coherent 2>&1 | cut -c1-330 | tee $OUT/acc_cfg3.log
echo "== cfg1/cfg3 shapes, throughput: D=200 b1 / D=400 b2, vocab 60238"
timeout 600 python bench.py --vocab 60238 --dim 200 --bitlevel 1 --tokens 17000000 --cpu-baseline none 2>/dev/null | short "cfg1 shape (V=60238 D=200 b1)"
timeout 600 python bench.py --vocab 60238 --dim 400 --bitlevel 2 --tokens 17000000 --cpu-baseline none 2>/dev/null | short "cfg3 shape (V=60238 D=400 b2)"
echo "== done"
