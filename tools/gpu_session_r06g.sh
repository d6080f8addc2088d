#!/bin/bash
# round 6, session g: host (background): SECOND reference runs for the three bands round 6 recorded once (heldout_v1m, long_d200,
# long_d400b2), so that they get a spread; GPU: tests/experiments/replicas8_cfg3.py on both streams at the command line's interval
# and at 1 M words (with the link-cost line), the exchange rule at two bits (configs[2] row length) on the long stream, then the
# whole -m gpu suite on the final tree.
set +e
export TMPDIR=/tmp
OUT=gpurun_out/r06g
mkdir -p $OUT /tmp/w2b_bands_a /tmp/w2b_bands_b
python - <<'PY'
import sys
sys.path.insert(0, "tests")
from w2b_testlib import write_heldout_corpus, write_headline_corpus
write_heldout_corpus("/tmp/long.txt", "long_d200")
write_headline_corpus("/tmp/headline.txt"); write_headline_corpus("/tmp/cfg1_100m.txt", n_zipf=98_000_000)
PY
( python tests/golden/make_fidelity_bands.py --out $OUT/bands_long_d200_run2.json --jobs long_d200 --heldout-big 256x1 --reuse-corpus /tmp/long.txt --tmp /tmp/w2b_bands_a > $OUT/bands_long_d200_run2.log 2>&1;
  python tests/golden/make_fidelity_bands.py --out $OUT/bands_long_d400b2_run2.json --jobs long_d400b2 --heldout-big 256x1 --reuse-corpus /tmp/long.txt --tmp /tmp/w2b_bands_a > $OUT/bands_long_d400b2_run2.log 2>&1;
  python tests/golden/make_fidelity_bands.py --out $OUT/bands_v1m_run2.json --jobs heldout_v1m --heldout-big 256x1 --tmp /tmp/w2b_bands_b > $OUT/bands_v1m_run2.log 2>&1;
  echo "reference bands done" ) &
REF=$!
echo "== (1) replicas8_cfg3.py"
timeout 900 python tests/experiments/replicas8_cfg3.py /tmp/headline.txt --out $OUT/replicas8_proxy.json 2>&1 | grep R8 | tee $OUT/replicas8_proxy.txt
timeout 900 python tests/experiments/replicas8_cfg3.py /tmp/cfg1_100m.txt --out $OUT/replicas8_literal.json 2>&1 | grep R8 | tee $OUT/replicas8_literal.txt
timeout 900 python tests/experiments/replicas8_cfg3.py /tmp/cfg1_100m.txt --positions 8192 --out $OUT/replicas8_literal_1m.json 2>&1 | grep R8 | tee $OUT/replicas8_literal_1m.txt
rm -f /tmp/headline.txt /tmp/cfg1_100m.txt
echo "== (2) the rule at two bits: size 400, bitlevel 2, 8 replicas x 32 workers on the 100 M-token long stream (-sample 0 here)"
RR="timeout 1500 python tests/experiments/replica_rules.py"
$RR /tmp/long.txt --size 400 --bitlevel 2 --workers 256 --positions 12288 --rules "lib2;lib2" --out $OUT/rules_b2_p12288.json 2>&1 | grep RR | tee $OUT/rules_b2_p12288.txt
$RR /tmp/long.txt --size 400 --bitlevel 2 --workers 256 --positions 6144 --rules "lib2" --out $OUT/rules_b2_p6144.json 2>&1 | grep RR | tee $OUT/rules_b2_p6144.txt
$RR /tmp/long.txt --size 400 --bitlevel 0 --workers 256 --positions 12288 --rules "lib2" --out $OUT/rules_b0_p12288.json 2>&1 | grep RR | tee $OUT/rules_b0_p12288.txt
echo "== (3) pytest -m gpu"
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
echo "== waiting for the host's reference runs"
wait $REF
rm -f /tmp/long.txt
tail -1 $OUT/bands_long_d200_run2.log $OUT/bands_long_d400b2_run2.log $OUT/bands_v1m_run2.log
echo "== done"
