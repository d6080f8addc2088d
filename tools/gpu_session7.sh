#!/bin/bash
set +e
OUT=gpurun_out/s7
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest worker"; timeout 900 python -m pytest tests/test_gpu_worker.py tests/test_gpu_cli.py -m gpu -q --no-header -p no:cacheprovider --tb=short -x 2>&1 | grep -E "^E  |^tests/|passed|failed|Error" | cut -c1-260 | tee $OUT/pytest.log
B="python bench.py --cpu-baseline none --also-relaxed 0 --form worker --tokens 50000000 --steps 12 --warmup 3"
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-40s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
for wc in 1 0; do for rl in 0 1; do
  timeout 600 $B --window-cache $wc --relaxed $rl 2>$OUT/err.log | short "worker zipf wc=$wc relaxed=$rl" | tee -a $OUT/variants.log
done; done
timeout 600 $B --window-cache 1 --ids uniform 2>>$OUT/err.log | short "worker uniform wc=1 coherent" | tee -a $OUT/variants.log
timeout 600 $B --window-cache 1 --workers 512 --positions 2048 2>>$OUT/err.log | short "worker zipf wc=1 workers=512" | tee -a $OUT/variants.log
timeout 600 $B --window-cache 1 --workers 2048 --positions 512 2>>$OUT/err.log | short "worker zipf wc=1 workers=2048" | tee -a $OUT/variants.log
tail -3 $OUT/err.log | cut -c1-300
echo "== loss fidelity (planted corpus)"
python - <<'PY' 2>&1 | tee $OUT/fidelity.log
import os, sys, subprocess
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from planted import make_planted
os.makedirs('/tmp/w2b_acc', exist_ok=True)
make_planted('/tmp/w2b_acc/planted.txt', '/tmp/w2b_acc/questions.txt')
for wc in (True, False):
    for th in (1, 8, 64, 256, 1024):
        code = ("import sys; sys.path.insert(0,'.'); import word2bits_amd as w; l=w.train_model('/tmp/w2b_acc/planted.txt','/tmp/w2b_acc/o.bin',"
                "bitlevel=1,size=200,window=8,negative=24,threads=%d,iter=5,min_count=5,binary=1,positions_per_launch=%d,window_cache=%r); print(l[-1])" % (th, 65536 if th < 64 else 4096, wc))
        p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
        q = subprocess.run("oracle/_ref/compute_accuracy /tmp/w2b_acc/o.bin 0 0 < /tmp/w2b_acc/questions.txt | grep 'Total accuracy' | tail -1", shell=True, capture_output=True, text=True)
        print("window_cache", wc, "threads", th, "last epoch loss", p.stdout.strip()[-30:], "|", q.stdout.strip()[:40], p.stderr[-300:] if p.returncode else "")
PY
echo "== done"
