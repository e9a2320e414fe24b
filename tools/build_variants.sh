#!/bin/bash
# Build container: phase-ablation / experiment builds of libssamd (never shipped as the product):
#   tools/build_variants.sh name1:-DFLAG1,-DFLAG2 name2:-DFLAG ...   -> tools/_exp/libssamd_<name>.so
# Select one at run time with SSAMD_LIB=<path> SSAMD_EXPERIMENT=1 (the binding refuses SSAMD_LIB alone).  tools/_exp/ is
# listed in .gpurunignore: remove that line for a session that runs ablations on the GPU box.
cd "$(dirname "$0")/.." || exit 1
mkdir -p tools/_exp
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fno-slp-vectorize -Wno-int-to-pointer-cast -DSSAMD_SINGLE_TU ${flags//,/ } \
    -o tools/_exp/libssamd_$name.so simplestereo_amd/csrc/ssamd_api.hip &
done
wait
ls -la tools/_exp/
