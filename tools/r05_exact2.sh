#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_exact2; mkdir -p $O
cd $R
timeout 300 python tools/time_exact.py 2>&1 | grep -v amdgpu | cut -c1-800 | tee $O/time_exact.txt | head -3
timeout 900 python -m pytest tests/test_gpu_exact.py tests/test_gpu_asw.py tests/test_gpu_full_frame.py -m gpu -q 2>&1 | tail -2
