#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python command; prints per-kernel averages.  usage: tools/r06_trace.sh <tag> <script.py> [args]
tag=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $O -o s -- python $R/"$@" > $O/out.log 2>&1
grep -E "1080p|case" $O/out.log | cut -c1-300
python - "$O" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-90s calls=%-4s avg_us=%10.1f max_us=%10.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
