#!/bin/bash
# round 5, first GPU call: the tree as it stands -- gpu tests, the bench line with the reworked cpu_baseline, and a
# >= 20-launch rocprofv3 kernel-stats pass of the bench command (the r04 average rested on 6 launches with one outlier)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_first; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o s -- python $R/bench.py --steps 24 --warmup 2 --no-cpu-baseline --no-others --no-e2e --no-bad1 > $O/stats.log 2>&1
python $R/tools/prof_summary.py $O > /dev/null; head -12 $O/summary.txt
