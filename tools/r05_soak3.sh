#!/bin/bash
# end-of-round soaks on the three-unit library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_soak3; mkdir -p $O
cd $R
for seed in 101 202 303; do SOAK_SEED=$seed timeout 500 python tools/soak_r05.py 360 2>&1 | grep -v amdgpu | tail -3 | cut -c1-300 | tee -a $O/soak_r05.txt; done
timeout 400 python tools/soak_pipe.py 240 2>&1 | grep -v amdgpu | tail -1 | tee $O/soak_pipe.txt
timeout 400 python tools/soak_wave.py 240 2>&1 | grep -v amdgpu | tail -1 | tee $O/soak_wave.txt
timeout 300 python tools/soak_threads.py 120 2>&1 | grep -v amdgpu | tail -1 | tee $O/soak_threads.txt
