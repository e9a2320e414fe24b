#!/bin/bash
# round 6, second GPU call: exact mode with the raw queue + filter (merging calls) -- parity tests, per-kernel times of the pass
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_second; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o s -- python $R/tools/time_exact.py > $O/time_exact.txt 2>&1
cut -c1-900 $O/time_exact.txt | grep case
cd $R && timeout 300 python tools/exact_raw_audit.py 2>&1 | tail -12; cd /tmp
python - "$O" <<'PY'
import csv, glob, os, sys
O = sys.argv[1]
for f in glob.glob(os.path.join(O, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print("%-100s calls=%-4s avg_us=%9.1f" % (row["Name"][:100], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
