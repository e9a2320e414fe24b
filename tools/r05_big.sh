#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05_big
timeout 600 python tools/exact_big.py 2>&1 | grep -v amdgpu | tee gpurun_out/r05_big/exact_big.txt
