#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_tests; mkdir -p $O
cd $R
timeout 2000 python -m pytest tests -m gpu -q -rA --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|FAILED|ERROR" $O/pytest.log | tail -30
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
