#!/bin/bash
# GPU box: VALU / LDS instruction counters + kernel stats of ONE configuration's dominant kernel, as the JSON bench.py replays
# (profiles/valu_<config>.json).  usage: tools/prof_valu.sh <config-name> <kernel-substring> [run_asw.py args...]
#   -> gpurun_out/valu_<config-name>.json, gpurun_out/profvalu_<config-name>/{stats,pmc}/
name=$1; kern=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profvalu_$name; mkdir -p $O
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o s -- python $R/tools/run_asw.py --steps 4 "$@" > $O/stats.log 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $O/pmc -o p -- python $R/tools/run_asw.py --steps 2 "$@" > $O/pmc.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/pmc2 -o p -- python $R/tools/run_asw.py --steps 2 "$@" > $O/pmc2.log 2>&1
python - "$O" "$name" "$kern" "$R" "$*" <<'PY'
import csv, glob, json, os, sys, collections
O, name, kern, R, args = sys.argv[1:6]
out = {"config": name, "run_asw_args": args}
for sub in ("pmc", "pmc2"):
    for f in glob.glob(os.path.join(O, sub, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter(); kname = None
        for row in csv.DictReader(open(f)):
            if kern in row["Kernel_Name"]:
                kname = row["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
        # per CALL of the operator: GSW launches the kernel twice per call (both passes): counts per launch x launches per step
        for c in acc:
            out[c + "_per_launch"] = acc[c] / n[c]
        if kname: out["kernel"] = kname
for f in glob.glob(os.path.join(O, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if kern in row["Name"]:
            out["kernel_stats"] = {"calls": int(row["Calls"]), "average_ns": float(row["AverageNs"]), "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"])}
out["plain_fp32_issue_peak_G_wave_instr_per_s_per_simd"] = 0.96
out["peak_source"] = "tools/ubench_vgpr_banks.hip, profiles/r03_ubench_vgpr_banks.txt: the tap mix at 4 waves per SIMD, 0.959 G wave-instr/s/SIMD"
out["source"] = "tools/prof_valu.sh %s %s %s (rocprofv3 --pmc SQ_INSTS_VALU ... and --kernel-trace --stats, separate passes)" % (name, kern, args)
json.dump(out, open(os.path.join(R, "gpurun_out", "valu_%s.json" % name), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
