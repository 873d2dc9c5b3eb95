#!/bin/bash
# Register / scratch summary of every kernel of one .hip source (cross-compiles; no GPU needed):
#   scripts/kres.sh dflash_amd/csrc/gemm_skinny.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Rpass-analysis=kernel-resource-usage "$@" -c "$src" -o /dev/null 2>&1 |
  awk '/Function Name:/{n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n)} /    VGPRs: /{v=$0; sub(/.*VGPRs: /,"",v); sub(/ \[.*/,"",v)} /ScratchSize/{s=$0; sub(/.*: /,"",s); sub(/ \[.*/,"",s)} /VGPRs Spill/{p=$0; sub(/.*: /,"",p); sub(/ \[.*/,"",p); printf "%-80s vgpr %-4s scratch %-4s spill %s\n", n, v, s, p}'
