#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05_smoke
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v amdgpu | tail -2
