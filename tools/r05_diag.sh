#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_diag; mkdir -p $O
cd $R
timeout 600 python tools/exact_diag.py > $O/exact_diag.txt 2>&1; cat $O/exact_diag.txt | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_exact.py tests/test_gpu_rows2.py tests/test_gpu_strips_multiprocess.py tests/test_gpu_asw.py -m gpu -q -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
