#!/bin/bash
set +e
OUT=gpurun_out/s5
mkdir -p $OUT
export TMPDIR=/tmp
echo "== coherence2"; timeout 300 ./tools/coherence_probe2 2>&1 | grep -E "G= 64|fg|uc" | tee $OUT/coherence2.log
export W2B_LIB=$PWD/word2bits_amd/libword2bits_hip_exp.so
B="python bench.py --cpu-baseline none --tokens 30000000 --steps 10 --warmup 2"
short() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-34s %8.2f Mw/s  frac %.3f  launch %.2f ms' % (sys.argv[1], d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms']))
" "$1"; }
for m in 0 1 2 3; do
  W2B_MEM_MODE=$m timeout 600 $B 2>/dev/null | short "tuples zipf memmode=$m" | tee -a $OUT/variants.log
  W2B_MEM_MODE=$m timeout 600 $B --form worker 2>/dev/null | short "worker zipf memmode=$m" | tee -a $OUT/variants.log
done
echo "== loss fidelity per mem mode (planted corpus, 8 and 256 workers)"
python - <<'PY' 2>&1 | tee $OUT/fidelity.log
import os, sys, subprocess, json
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from planted import make_planted
os.makedirs('/tmp/w2b_acc', exist_ok=True)
make_planted('/tmp/w2b_acc/planted.txt', '/tmp/w2b_acc/questions.txt')
for m in (0, 1, 2, 3):
    for th in (8, 256):
        code = ("import sys; sys.path.insert(0,'.'); import word2bits_amd as w; l=w.train_model('/tmp/w2b_acc/planted.txt','/tmp/w2b_acc/o.bin',"
                "bitlevel=1,size=200,window=8,negative=24,threads=%d,iter=5,min_count=5,binary=1,positions_per_launch=4096); print(l[-1])" % th)
        p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=dict(os.environ, W2B_MEM_MODE=str(m)))
        print("memmode", m, "threads", th, "last epoch loss", p.stdout.strip()[-40:], p.stderr[-200:] if p.returncode else "")
PY
unset W2B_LIB
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --tb=short 2>&1 | grep -E "^E  |^tests/|passed|failed|Error" | cut -c1-260 | tee $OUT/pytest.log
echo "== done"
