#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_strip; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_rows2.py tests/test_gpu_strips_multiprocess.py tests/test_gpu_rccl_world1.py tests/test_gpu_asw.py -m gpu -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log

python tools/strip_overlap_cost.py > $O/strip_overlap_cost.txt 2>&1; grep -v amdgpu $O/strip_overlap_cost.txt
