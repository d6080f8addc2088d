#!/bin/bash
# round 6, session f: (1) does the exchange rule carry to regimes it was not found on?  8 replicas on the text8-sized corpus (size 200,
# 32 workers each, against 256 workers in one replica) and on heldout_v1m (V = 1 M, size 512, window 5, negative 10; 128 workers each
# against 1024); (2) the whole -m gpu suite on the final tree.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r06f
mkdir -p $OUT
python - <<'PY'
import sys
sys.path.insert(0, "tests")
from w2b_testlib import write_zipf_text_corpus, write_heldout_corpus
write_zipf_text_corpus("/tmp/t8.txt"); write_heldout_corpus("/tmp/v1m.txt", "heldout_v1m")
PY
RR="timeout 1500 python tests/experiments/replica_rules.py"
echo "== (1a) text8-sized corpus, 8 replicas x 32 workers"
$RR /tmp/t8.txt --size 200 --workers 256 --positions 2048 --rules "lib2;lib2" --out $OUT/rules_t8_p2048.json 2>&1 | grep RR | tee $OUT/rules_t8_p2048.txt
$RR /tmp/t8.txt --size 200 --workers 256 --positions 4096 --rules "lib2" --out $OUT/rules_t8_p4096.json 2>&1 | grep RR | tee $OUT/rules_t8_p4096.txt
$RR /tmp/t8.txt --size 200 --workers 256 --positions 1024 --rules "lib2" --out $OUT/rules_t8_p1024.json 2>&1 | grep RR | tee $OUT/rules_t8_p1024.txt
echo "== (1b) heldout_v1m, 8 replicas x 128 workers"
$RR /tmp/v1m.txt --size 512 --window 5 --negative 10 --positions 3456 --rules "lib2;lib2" --out $OUT/rules_v1m_p3456.json 2>&1 | grep RR | tee $OUT/rules_v1m_p3456.txt
$RR /tmp/v1m.txt --size 512 --window 5 --negative 10 --positions 8192 --rules "lib2" --out $OUT/rules_v1m_p8192.json 2>&1 | grep RR | tee $OUT/rules_v1m_p8192.txt
$RR /tmp/v1m.txt --size 512 --window 5 --negative 10 --positions 1728 --rules "lib2" --out $OUT/rules_v1m_p1728.json 2>&1 | grep RR | tee $OUT/rules_v1m_p1728.txt
rm -f /tmp/t8.txt /tmp/v1m.txt
echo "== (2) pytest -m gpu"
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -30 | tee $OUT/pytest_gpu.txt
echo "== done"
