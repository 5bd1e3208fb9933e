#!/bin/bash
# tools/kernel_resources.sh <file.hip> [-Dflags...]: VGPRs / scratch / SGPR spills / occupancy of every kernel of one source file
# (hipcc -Rpass-analysis=kernel-resource-usage, compact).  Development aid; no GPU needed.
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I"$(dirname "$0")/../vnext_amd/csrc" "$@" -c -o /dev/null "$f" \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|    VGPRs:|ScratchSize|Occupancy|SGPRs Spill" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' \
  | paste - - - - - | sed 's/Function Name: //' | while read -r name rest; do echo "$(echo "$name" | c++filt | cut -c1-110) | $rest"; done
