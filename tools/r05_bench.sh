#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_bench; mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 300 $O/bench.err
timeout 300 python tools/bench_pointwise.py 2>/dev/null | tail -1 > $O/pointwise.json
timeout 600 python bench.py --steps 10 --warmup 3 --consistent --no-others --no-e2e --no-cpu-baseline > $O/bench_consistent.json 2> $O/bench_consistent.err; echo "bench consistent rc=$?"
