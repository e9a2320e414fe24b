#!/bin/bash
# round 5, profiling call: >= 20 profiled launches of the headline kernel (rocprofv3 kernel stats of the bench command), the
# HBM-traffic and VALU-instruction counter passes of config 3 (separate --pmc runs), and the exact-mode kernels' share
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_third; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python tools/exact_audit.py > $O/exact_audit.txt 2>&1; cat $O/exact_audit.txt | tail -3
timeout 300 python tools/time_exact.py > $O/time_exact.txt 2>&1; cut -c1-700 $O/time_exact.txt
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-others --no-e2e --no-bad1"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o s -- $BENCH --steps 24 --warmup 2 > $O/stats.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- $BENCH --steps 4 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o p -- $BENCH --steps 4 --warmup 1 > $O/pmc_write.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $O/pmc_valu -o p -- $BENCH --steps 4 --warmup 1 > $O/pmc_valu.log 2>&1
python $R/tools/prof_summary.py $O > /dev/null
python - "$O" <<'PY'
import csv, glob, json, os, sys, collections
O = sys.argv[1]
out = {}
for name, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in glob.glob(os.path.join(O, name, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == key:
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[k] += float(row["Counter_Value"]); n[k] += 1
        for k in acc:
            out.setdefault(k, {})[key + "_KB_per_launch"] = acc[k] / n[k]
            out[k]["launches_" + key] = n[k]
for k, v in out.items():
    if "FETCH_SIZE_KB_per_launch" in v and "WRITE_SIZE_KB_per_launch" in v:
        v["hbm_bytes_per_launch"] = (2 * v["FETCH_SIZE_KB_per_launch"] + v["WRITE_SIZE_KB_per_launch"]) * 1024
json.dump(out, open(os.path.join(O, "traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in out.items() if "ssamd" in k}, indent=1))
PY
head -14 $O/summary.txt
cd $R
# exact mode under the profiler: which of its kernels costs what (config 3 frame)
cd /tmp; rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_exact -o s -- python $R/tools/time_exact.py > $O/stats_exact.log 2>&1
python - "$O" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "stats_exact", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print("%-70s calls=%s avg_ns=%s total_ns=%s" % (row["Name"][:70], row["Calls"], row["AverageNs"], row["TotalDurationNs"]))
PY

cd $R; timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
