#!/bin/bash
# GPU box: rocprofv3 evidence for the bench command (kernel stats + HBM traffic counters).
# usage: tools/prof_bench.sh <tag>   -> gpurun_out/profbench_<tag>/{stats/,pmc_fetch/,pmc_write/,summary.txt,traffic.json}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profbench_$tag; mkdir -p $O
CMD="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-others --no-e2e --no-bad1 $@"
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o s -- $CMD > $O/stats.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- $CMD > $O/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o p -- $CMD > $O/pmc_write.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $O/pmc_valu -o p -- $CMD > $O/pmc_valu.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc_wave -o p -- $CMD > $O/pmc_wave.log 2>&1
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
for k, v in out.items():
    if "FETCH_SIZE_KB_per_launch" in v and "WRITE_SIZE_KB_per_launch" in v:
        # gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section)
        v["hbm_bytes_per_launch"] = (2 * v["FETCH_SIZE_KB_per_launch"] + v["WRITE_SIZE_KB_per_launch"]) * 1024
json.dump(out, open(os.path.join(O, "traffic.json"), "w"), indent=1, sort_keys=True)
# VALU / LDS instruction counts per launch of every kernel (pmc_valu pass)
cnt = {}
for f in glob.glob(os.path.join(O, "pmc_valu", "**", "*counter_collection.csv"), recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
    for k in acc:
        cnt[k] = {c + "_per_launch": acc[k][c] / n[k][c] for c in acc[k]}
json.dump(cnt, open(os.path.join(O, "valu.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: v for k, v in out.items() if "ssamd" in k}, indent=1))
PY
grep -A8 "kernel stats" $O/summary.txt | head -10
