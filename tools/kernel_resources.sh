#!/bin/bash
# VGPRs / SGPR spills / scratch / occupancy of every kernel of one translation unit of libssamd (hipcc -Rpass-analysis=kernel-resource-usage).
#   tools/kernel_resources.sh asw_pipe_tu.hip [extra hipcc flags]      (flags of simplestereo_amd/build.py UNITS are NOT added automatically)
cd "$(dirname "$0")/../simplestereo_amd/csrc" || exit 1
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-int-to-pointer-cast "$@" \
    -Rpass-analysis=kernel-resource-usage -c -o /tmp/kres_$$.o "$src" 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
       /    VGPRs: /{v=$0; sub(/.*VGPRs: /,"",v); sub(/ \[.*/,"",v)}
       /SGPRs: /{sg=$0; sub(/.*SGPRs: /,"",sg); sub(/ \[.*/,"",sg)}
       /ScratchSize/{sc=$0; sub(/.*: /,"",sc); sub(/ \[.*/,"",sc)}
       /Occupancy/{oc=$0; sub(/.*: /,"",oc); sub(/ \[.*/,"",oc)}
       /SGPRs Spill/{ss=$0; sub(/.*: /,"",ss); sub(/ \[.*/,"",ss)}
       /VGPRs Spill/{vs=$0; sub(/.*: /,"",vs); sub(/ \[.*/,"",vs)}
       /LDS Size/{printf "%-110s VGPRs %s SGPRs %s scratch %s occ %s sgpr_spill %s vgpr_spill %s\n", name, v, sg, sc, oc, ss, vs}'
rm -f /tmp/kres_$$.o
