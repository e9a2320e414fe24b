#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_prim; mkdir -p $O
cd $R
SOAK_SEED=5 timeout 500 python tools/soak_r05.py 300 > $O/soak_r05_a.txt 2>&1; grep -v amdgpu $O/soak_r05_a.txt | cut -c1-300 | tail -6
SOAK_SEED=77 timeout 500 python tools/soak_r05.py 300 > $O/soak_r05_b.txt 2>&1; grep -v amdgpu $O/soak_r05_b.txt | cut -c1-300 | tail -6
