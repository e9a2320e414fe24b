#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_exactbits; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_exact.py tests/test_gpu_asw.py tests/test_gpu_full_frame.py tests/test_gpu_photo_golden.py tests/test_gpu_wide_golden.py -m gpu -q -rA > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|^FAILED|^ERROR|exact:|identical to the reference|pixels differing|non-tie|full frame" $O/pytest.log | cut -c1-220 | tail -70
timeout 300 python tools/exact_diag.py 2>&1 | grep -v amdgpu | cut -c1-300 > $O/exact_diag.txt; cat $O/exact_diag.txt
timeout 300 python tools/time_exact.py 2>&1 | grep -v amdgpu | cut -c1-900 > $O/time_exact.txt; cat $O/time_exact.txt
timeout 300 python tools/bench_pointwise.py > $O/pointwise.txt 2>&1; tail -5 $O/pointwise.txt
