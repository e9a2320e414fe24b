"""summarise rocprofv3 outputs of tools/prof.sh into one text file (kernel stats + PMC sums per kernel)"""
import csv, glob, os, sys, collections
O = sys.argv[1]
lines = []
for f in glob.glob(os.path.join(O, "stats", "**", "*kernel_stats.csv"), recursive=True):
    lines.append("== kernel stats (%s)" % os.path.relpath(f, O))
    for row in csv.DictReader(open(f)):
        lines.append("  %-60s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row.get("Name", "")[:60], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
for p in sorted(glob.glob(os.path.join(O, "pmc*"))):
    if not os.path.isdir(p): continue
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); meta = {}
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:50]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            meta[k] = "vgpr=%s sgpr=%s lds=%s wg=%s grid=%s" % (row.get("VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"), row.get("Workgroup_Size"), row.get("Grid_Size"))
        lines.append("== %s" % os.path.relpath(f, O))
        for k in acc:
            lines.append("  %s  [%s]" % (k, meta[k]))
            for c, v in sorted(acc[k].items()):
                lines.append("      %-28s %.6g" % (c, v))
open(os.path.join(O, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
