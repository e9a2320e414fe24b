#!/bin/bash
# round 6, profiling call: the bench line (default path = exact) + its details, >= 24 profiled launches of the headline kernel for the
# default path and for --fp32 (rocprofv3 kernel stats of the bench command), the HBM-traffic and VALU-instruction counter passes
# (separate --pmc runs), the exact pass kernel by kernel, the loopback strip step (host cost, exchange hidden or not)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_prof; mkdir -p $O
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$? bytes=$(wc -c < $O/bench_line.json)"
cp bench_details.json $O/bench_details.json
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --no-others --no-e2e --no-bad1"
timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o s -- $BENCH --steps 24 --warmup 2 > $O/stats.log 2>&1
timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_fp32 -o s -- $BENCH --fp32 --steps 24 --warmup 2 > $O/stats_fp32.log 2>&1
timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats_cons -o s -- $BENCH --consistent --steps 12 --warmup 2 > $O/stats_cons.log 2>&1
timeout 400 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- $BENCH --steps 4 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o p -- $BENCH --steps 4 --warmup 1 > $O/pmc_write.log 2>&1
timeout 400 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $O/pmc_valu -o p -- $BENCH --steps 4 --warmup 1 > $O/pmc_valu.log 2>&1
python - "$O" <<'PY'
import csv, glob, json, os, sys, collections
O = sys.argv[1]
res = {}
for tag in ("stats", "stats_fp32", "stats_cons"):
    rows = []
    for f in glob.glob(os.path.join(O, tag, "**", "*kernel_stats.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    res[tag] = [{"name": r["Name"], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                 "max_us": float(r["MaxNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6} for r in rows]
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
        v["hbm_bytes_per_launch"] = (2 * v["FETCH_SIZE_KB_per_launch"] + v["WRITE_SIZE_KB_per_launch"]) * 1024      # gfx950 correction (MI355X_MICROARCH.md)
cnt = {}
for f in glob.glob(os.path.join(O, "pmc_valu", "**", "*counter_collection.csv"), recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
    for k in acc:
        cnt[k] = {c + "_per_launch": acc[k][c] / n[k][c] for c in acc[k]}
        cnt[k]["launches_counted"] = max(n[k].values())
json.dump({"kernel_stats": res, "traffic": out, "valu": cnt}, open(os.path.join(O, "summary.json"), "w"), indent=1, sort_keys=True)
for tag in res:
    print("==", tag)
    for r in sorted(res[tag], key=lambda r: -r["total_ms"])[:9]:
        print("  %-96s calls=%-4d avg_us=%10.1f min=%10.1f max=%10.1f" % (r["name"][:96], r["calls"], r["avg_us"], r["min_us"], r["max_us"]))
print(json.dumps({k: v for k, v in out.items() if "ssamd" in k}, indent=1))
print(json.dumps({k: v for k, v in cnt.items() if "aggregate" in k}, indent=1))
PY
cd $R
timeout 600 python tools/time_exact.py > $O/time_exact.txt 2>&1
for p in 1 0; do TORCH_NCCL_HIGH_PRIORITY=$p timeout 600 python tools/strip_host_cost.py 8 3 2>&1 | grep -E "case|PRIORITY"; done > $O/strip_host_cost.txt
timeout 300 python tools/strip_host_cost.py 4 1 2>&1 | grep -E "case" >> $O/strip_host_cost.txt
timeout 600 python tools/exact_big.py > $O/exact_big.txt 2>&1
tail -3 $O/exact_big.txt; cat $O/bench_line.json
