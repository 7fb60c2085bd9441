#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output: per-kernel time (kernel_stats) and PMC
counter sums per kernel (counter_collection)."""
import csv, sys, collections, glob, os
d = sys.argv[1]
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    print("==", f)
    for row in csv.DictReader(open(f)):
        print("  %-60s calls=%s total_ms=%.3f avg_ms=%.3f pct=%s" % (row["Name"][:60], row["Calls"], float(row["TotalDurationNs"]) / 1e6, float(row["AverageNs"]) / 1e6, row["Percentage"]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    print("==", f)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    meta = {}
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:50]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        meta[k] = (row.get("VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"), row.get("Scratch_Size"), row.get("Grid_Size"), row.get("Workgroup_Size"))
    for k, v in acc.items():
        if "k_reg" in k or "lds_eval" in k or "k_hbm" in k or len(acc) < 4:
            print("  ", k, "vgpr/sgpr/lds/scratch/grid/wg=", meta[k])
            for c, val in sorted(v.items()):
                print("      %-28s %.4g" % (c, val))
