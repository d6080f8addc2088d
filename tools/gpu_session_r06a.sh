#!/bin/bash
# round 6, session a (first contact; the host runs the reference bands meanwhile):
#  host (background): second 256-thread reference run of BASELINE configs[1] literally (cfg1_100m, ~17 min) so that the band has a
#        sigma; then the NEW held-out full-device long-stream regime heldout_v1m (V = 1 M, Zipf 1.1, size 512, window 5,
#        negative 10, 80 M tokens), recorded before any constant of the full-device mode is touched;
#  GPU:  (1) coherence probe with the nt-load + plain-store cell; (2) the new oracle / refresher tests of the row-group kernel;
#        (3) same-box A/B of the hot-row copies' store policy -- nt (round 5: written through) against plain write-back
#        (libword2bits_hip_wb.so) -- throughput (bench.py headline only, alternating) and epoch loss on the 22 M-token proxy and
#        on the literal 100 M-token stream; (4) combination rules of the 8-replica exchange (tests/experiments/replica_rules.py).
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r06a
mkdir -p $OUT /tmp/w2b_bands_a /tmp/w2b_bands_b
R=$GRAFT_REPO_ROOT
( python tests/golden/make_fidelity_bands.py --out $OUT/bands_cfg1_run2.json --jobs cfg1_100m --cfg1 256x1 --tmp /tmp/w2b_bands_a > $OUT/bands_cfg1_run2.log 2>&1;
  python tests/golden/make_fidelity_bands.py --out $OUT/bands_v1m.json --jobs heldout_v1m --heldout-big 256x1 --tmp /tmp/w2b_bands_b > $OUT/bands_v1m.log 2>&1;
  echo "reference bands done" ) &
REF=$!
echo "== (1) coherence probe"; timeout 120 tools/coherence_probe2 2>&1 | grep -E "nt\+pl|^nt |sc1  " | tee $OUT/coherence_probe2.txt
echo "== (2) row-group kernel: oracle + refresher tests"
timeout 900 python -m pytest tests/test_gpu_groups.py -q -m gpu -x 2>&1 | tail -6 | tee $OUT/pytest_groups.txt
python - <<'PY'
import sys, time
sys.path.insert(0, "tests")
from w2b_testlib import write_headline_corpus
t = time.time(); write_headline_corpus("/tmp/headline.txt"); print("headline corpus written in %.0f s" % (time.time() - t), flush=True)
t = time.time(); write_headline_corpus("/tmp/cfg1_100m.txt", n_zipf=98_000_000); print("cfg1 corpus written in %.0f s" % (time.time() - t), flush=True)
PY
echo "== (3) store policy of the per-XCD copies: nt (A) vs plain write-back (B)"
B="python bench.py --steps 20 --warmup 5 --cpu-baseline none --cpu-cfg0 0 --also-relaxed 0 --also-legs 0 --also-shapes 0"
for i in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then L=$R/word2bits_amd/libword2bits_hip.so; else L=$R/word2bits_amd/libword2bits_hip_wb.so; fi
    W2B_LIB=$L timeout 300 $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('AB $v round $i: %.2f M words/s, %.3f ms/step, roofline %.4f' % (d['value']/1e6, d['ms_per_step'], r['frac']))" | tee -a $OUT/ab_store_policy.txt
  done
done
F="-bitlevel 1 -size 800 -window 8 -negative 24 -iter 1 -sample 0 -min-count 5 -binary 1 -threads 1024"
run() {   # variant, file, ref loss
  if [ $1 = A ]; then PRE=""; else PRE="$R/word2bits_amd/libword2bits_hip_wb.so"; fi
  LD_PRELOAD=$PRE ./word2bits -train $2 -output /dev/null $F > $OUT/run.txt 2> $OUT/run.err
  python - "$1" "$2" "$3" <<PY | tee -a $OUT/ab_store_policy.txt
import re, sys
out = open("$OUT/run.txt").read().replace("\r", "\n")
L = [float(x) for x in re.findall(r"Epoch Loss: (-?[\d.]+)", out)]
ref = float(sys.argv[3])
print("FID %s %-22s loss %.0f deviation %+.2f %% of the reference" % (sys.argv[1], sys.argv[2], L[0], 100 * (L[0] - ref) / abs(ref)))
PY
}
for v in A B A B; do run $v /tmp/headline.txt -126013238; done
for v in A B; do run $v /tmp/cfg1_100m.txt -543450078.458814; done
echo "== (4) replica exchange rules, 8 replicas x 128 workers, 22 M-token proxy"
RR="timeout 1500 python tests/experiments/replica_rules.py"
$RR /tmp/headline.txt --positions 1024 --rules "lib2;hard:32;smooth:8:8;smooth:16:16;smooth:32:32;smooth:64:64;smooth:128:128;smooth:256:256;smooth:32:128;smooth:128:32;agree:1;smooth:32:32,bf16" --out $OUT/rules_p1024.json 2>&1 | grep RR | tee $OUT/rules_p1024.txt
S1=$(python -c "import json; print(json.load(open('$OUT/rules_p1024.json'))['single_replica_loss'])")
$RR /tmp/headline.txt --positions 8192 --single $S1 --rules "lib2;smooth:16:16;smooth:32:32;smooth:64:64;smooth:128:128" --out $OUT/rules_p8192.json 2>&1 | grep RR | tee $OUT/rules_p8192.txt
$RR /tmp/headline.txt --positions 4096 --single $S1 --rules "smooth:32:32;smooth:64:64" --out $OUT/rules_p4096.json 2>&1 | grep RR | tee $OUT/rules_p4096.txt
$RR /tmp/headline.txt --positions 128 --single $S1 --rules "sum;smooth:32:32;smooth:64:64" --out $OUT/rules_p128.json 2>&1 | grep RR | tee $OUT/rules_p128.txt
$RR /tmp/headline.txt --positions 512 --sync geom:1:16 --single $S1 --rules "smooth:32:32;smooth:64:64" --out $OUT/rules_geom.json 2>&1 | grep RR | tee $OUT/rules_geom.txt
echo "== (4b) the literal 100 M-token stream"
$RR /tmp/cfg1_100m.txt --positions 8192 --rules "lib2;smooth:32:32;smooth:64:64" --out $OUT/rules_100m_p8192.json 2>&1 | grep RR | tee $OUT/rules_100m_p8192.txt
S2=$(python -c "import json; print(json.load(open('$OUT/rules_100m_p8192.json'))['single_replica_loss'])")
$RR /tmp/cfg1_100m.txt --positions 1024 --single $S2 --rules "smooth:32:32;smooth:64:64" --out $OUT/rules_100m_p1024.json 2>&1 | grep RR | tee $OUT/rules_100m_p1024.txt
rm -f /tmp/cfg1_100m.txt /tmp/headline.txt
echo "== waiting for the host's reference runs"
wait $REF
tail -3 $OUT/bands_cfg1_run2.log $OUT/bands_v1m.log
echo "== done"
