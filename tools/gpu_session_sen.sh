#!/bin/bash
# A/B of two builds of the library on ONE box (W2B_LIB selects the library; `make SUFFIX=_x [W2B_DEFS=...]` builds them).
# usage: tools/gpu_session_sen.sh LIB_A LIB_B [phase]   -- "phase": both libraries are -DW2B_PHASE_TIMERS builds, print
# the phase ticks of worker 0 (W2B_DEBUG) next to the rates
set +e
export TMPDIR=/tmp
mkdir -p gpurun_out
A=${1:-libword2bits_hip_old.so}; B=${2:-libword2bits_hip.so}; PH=$3
run() {
  W2B_DEBUG=${PH:+1} W2B_LIB=$PWD/word2bits_amd/$1 timeout 60 python bench.py "${@:2}" --tokens 30000000 --steps 10 --warmup 3 \
    --cpu-baseline none --also-relaxed 0 --also-legs 0 --also-shapes 0 2>gpurun_out/ab.err | grep '^{' | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f M words/s  frac %.3f  %.3f ms  workers %s  per_cu %s' % (d['value']/1e6, d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['config']['workers'], d['config']['worker_kernel']['workers_per_cu']))"
  [ -n "$PH" ] && grep "phase ticks" gpurun_out/ab.err | grep -v "\[10\]=0" | tail -1
}
for lib in $A $B $A $B; do
  echo "== $lib"
  echo -n "d200:          "; run $lib --vocab 60238 --dim 200
  [ -z "$PH" ] && { echo -n "d400 b2:       "; run $lib --vocab 60238 --dim 400 --bitlevel 2; }
  echo -n "cfg2 resident: "; run $lib --window-cache 1
done 2>&1 | tee gpurun_out/ab_${A%.so}_vs_${B%.so}.txt
