#!/bin/bash
# round 6, first GPU call: exact mode v2 (in-kernel near-tie selection) -- parity tests, cost, a bench line
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_exact.py tests/test_gpu_rows2.py tests/test_gpu_full_frame.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r06_first_tests.log
timeout 600 python tools/time_exact.py > gpurun_out/r06_time_exact.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err
cp bench_details.json gpurun_out/r06_bench_a_details.json 2>/dev/null
tail -5 gpurun_out/r06_first_tests.log; cat gpurun_out/r06_time_exact.log | cut -c1-600; wc -c gpurun_out/r06_bench_a.json; cat gpurun_out/r06_bench_a.json
