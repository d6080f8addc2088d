#!/bin/bash
# producer flags out of scratch memory (sentence-resident kernel): parity tests of the kernel, then A/B on one box
# against the previous build (W2B_LIB=..._old.so) at equal worker counts
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
W2B_TEST_NO_TORCH=1 timeout 100 python -m pytest tests/test_gpu_worker.py -m gpu -x -q --no-header -p no:cacheprovider --tb=short 2>&1 | tail -6 | cut -c1-300 | tee gpurun_out/flags_tests.txt
run() {
  W2B_LIB=$PWD/word2bits_amd/$1 timeout 60 python bench.py "${@:2}" --tokens 30000000 --steps 10 --warmup 3 \
    --cpu-baseline none --also-relaxed 0 --also-legs 0 --also-shapes 0 2>&1 | grep '^{' | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f M words/s  frac %.3f  %.3f ms  workers %s  per_cu %s' % (d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['config']['workers'], d['config']['worker_kernel']['workers_per_cu']))"
}
for lib in libword2bits_hip_old.so libword2bits_hip.so libword2bits_hip_old.so libword2bits_hip.so; do
  echo "== $lib"
  echo -n "d200:          "; run $lib --vocab 60238 --dim 200
  echo -n "d400 b2:       "; run $lib --vocab 60238 --dim 400 --bitlevel 2
  echo -n "cfg2 resident: "; run $lib --window-cache 1
done 2>&1 | tee gpurun_out/flags_ab.txt
