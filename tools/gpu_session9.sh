#!/bin/bash
set +e
export TMPDIR=/tmp
export W2B_LIB=$PWD/word2bits_amd/libword2bits_hip_dbg.so W2B_DEBUG=1
B="python bench.py --cpu-baseline none --also-relaxed 0 --tokens 30000000 --steps 8 --warmup 2"
timeout 600 $B 2>&1 | grep -E "w2b debug" | cut -c1-400
timeout 600 $B --workers 256 2>&1 | grep -E "w2b debug: phase" | cut -c1-400
