"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small text summary."""
import csv, glob, os, sys, collections
root = sys.argv[1]
def rows(pattern):
    for f in glob.glob(os.path.join(root, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r
print("# kernel stats (rocprofv3 --kernel-trace --stats)")
for f in glob.glob(os.path.join(root, "trace/**/*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for i, line in enumerate(fh):
            if i < 12: print(line.rstrip())
print("\n# per-kernel durations from kernel_trace (ns): count, mean, min, max")
dur = collections.defaultdict(list)
for r in rows("trace/**/*kernel_trace.csv"):
    dur[r["Kernel_Name"][:90]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    meta = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
    dur[r["Kernel_Name"][:90] + " :: vgpr,sgpr,lds,scratch,wg,grid"] = [meta]
for k, v in dur.items():
    if "::" in k: print(" ", k, v[0]); continue
    print("  %-92s n=%d mean=%.0f min=%d max=%d" % (k, len(v), sum(v) / len(v), min(v), max(v)))
print("\n# PMC counters: mean per dispatch, by kernel")
for p in ("pmc_sq", "pmc_sq2", "pmc_sq3", "pmc_sq4", "pmc_fetch", "pmc_write"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows(p + "/**/*counter_collection.csv"):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "ssw_" not in k and "chamfer" not in k: continue
        print("  [%s] %s" % (p, k))
        for c, v in sorted(d.items()):
            print("      %-28s mean=%.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
