#!/bin/bash
# usage: tools/kernel_resources.sh <file.hip> [filter]   -- SGPRs / VGPRs / spills / scratch / occupancy of every kernel in a translation unit
cd "$(dirname "$0")/../word2bits_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $W2B_DEFS -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|TotalSGPRs|VGPRs:|SGPRs Spill|VGPRs Spill|ScratchSize|Occupancy" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' \
 | awk '/Function Name/{if(n)print n r; n=$3; r=""; next}{gsub(/^ +/,""); r=r" | "$0}END{print n r}' | c++filt | sed 's/(anonymous namespace):://' | grep "${2:-.}"
