#!/bin/bash
# usage: tools/kernel_resources.sh pgx_scan.hip [name-filter]
# VGPR / SGPR / LDS / scratch / occupancy of every kernel of one translation unit (clang's
# -Rpass-analysis=kernel-resource-usage; cross-compiles, no GPU needed).
root=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I"$root/include" -I"$root/pygmu2_amd/csrc" \
  -Rpass-analysis=kernel-resource-usage -c "$root/pygmu2_amd/csrc/$1" -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size" |
  sed -E 's/.*remark: [^ ]+ //; s/\[-Rpass.*//' | paste - - - - - - | { if [ -n "$2" ]; then grep -E "$2"; else cat; fi; }
