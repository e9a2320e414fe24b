#!/usr/bin/env python3
"""Build container: copy the summaries of the last tools/r06_profiles.sh run (gpurun_out/r06_prof/) into profiles/ -- bench line + details,
rocprofv3 kernel stats (default path, --fp32, --consistent), the text summary, and the replayed-counter JSONs bench.py reads."""
import glob, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, D = os.path.join(ROOT, "gpurun_out", "r06_prof"), os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(P, "bench_line.json"), os.path.join(D, "r06_bench_line.json"))
shutil.copy(os.path.join(P, "bench_details.json"), os.path.join(D, "r06_bench_details.json"))
for tag, name in (("stats", "r06_bench_c3_kernel_stats.csv"), ("stats_fp32", "r06_bench_c3_fp32_kernel_stats.csv"), ("stats_cons", "r06_bench_c3_consistent_kernel_stats.csv")):
    f = glob.glob(os.path.join(P, tag, "**", "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(f, os.path.join(D, name))
S = json.load(open(os.path.join(P, "summary.json")))
lines = ["Round 6: rocprofv3 evidence for the bench command, FINAL library of the round (tools/r06_profiles.sh, one MI355X, one gpurun call).",
         "  python bench.py --no-cpu-baseline --no-others --no-e2e --no-bad1 --steps 24 --warmup 2          (default path = exact: 26 launches of the timed command",
         "                                                                                               + the alternating fp32 / default comparison loop)",
         "  ... --fp32 (StereoASW(exact=False): the fp32 argmin alone), ... --consistent --steps 12",
         "rocprofv3 --kernel-trace --stats; counters in SEPARATE passes (--pmc FETCH_SIZE / WRITE_SIZE / SQ_*), --steps 4 --warmup 1.",
         "HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (gfx950 correction of MI355X_MICROARCH.md).", ""]
for tag, lab in (("stats", "default path (exact)"), ("stats_fp32", "--fp32"), ("stats_cons", "--consistent (default path)")):
    lines.append("== kernel stats, %s" % lab)
    for r in sorted(S["kernel_stats"][tag], key=lambda r: -r["total_ms"]):
        if r["total_ms"] >= 0.01:
            lines.append("  %-100s calls=%-4d avg_us=%10.1f min_us=%10.1f max_us=%10.1f total_ms=%9.2f" % (r["name"][:100], r["calls"], r["avg_us"], r["min_us"], r["max_us"], r["total_ms"]))
    lines.append("")
lines.append("== HBM traffic per launch (PMC), default path")
for k, v in S["traffic"].items():
    if "ssamd" in k and "hbm_bytes_per_launch" in v:
        lines.append("  %-70s FETCH_SIZE %.1f KB  WRITE_SIZE %.1f KB  -> %.1f MB per launch (%d launches)" % (k[:70], v["FETCH_SIZE_KB_per_launch"], v["WRITE_SIZE_KB_per_launch"], v["hbm_bytes_per_launch"] / 1e6, v["launches_FETCH_SIZE"]))
lines += ["", "== instruction counters per launch (PMC), default path"]
for k, v in S["valu"].items():
    if "ssamd" in k:
        lines.append("  %-70s %s" % (k[:70], "  ".join("%s=%.4g" % (c.replace("_per_launch", ""), x) for c, x in sorted(v.items()))))
open(os.path.join(D, "r06_bench_c3_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
agg = [k for k in S["traffic"] if "asw_aggregate" in k][0]
tp = os.path.join(D, "traffic_c3_1080p_d192_w35.json")
t = json.load(open(tp))
new = {"source": t["source"]}
for k, v in S["traffic"].items():
    if "ssamd" in k:
        new[k] = v
new["round5_values"] = t.get("round5_values", {})
json.dump(new, open(tp, "w"), indent=1)
vp = os.path.join(D, "valu_c3_1080p_d192_w35.json")
v = json.load(open(vp))
c = S["valu"][agg]
for k in ("SQ_INSTS_VALU_per_launch", "SQ_INSTS_LDS_per_launch", "SQ_LDS_IDX_ACTIVE_per_launch", "SQ_LDS_BANK_CONFLICT_per_launch", "SQ_WAVES_per_launch"):
    v[k] = c[k]
v["launches_counted"] = c["launches_counted"]
v["kernel"] = agg
ks = [r for r in S["kernel_stats"]["stats"] if "asw_aggregate" in r["name"]][0]
v["kernel_stats"] = {"calls": ks["calls"], "average_ns": ks["avg_us"] * 1e3, "min_ns": ks["min_us"] * 1e3, "max_ns": ks["max_us"] * 1e3}
json.dump(v, open(vp, "w"), indent=1)
b = json.load(open(os.path.join(P, "bench_line.json")))
print("bench", b["ms_per_step"], "kernel", b["roofline"]["kernel_ms"], "frac", b["roofline"]["frac"], "VALU", c["SQ_INSTS_VALU_per_launch"], "agg avg us", ks["avg_us"])
