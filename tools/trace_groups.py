#!/usr/bin/env python3
"""Groups the launches of a rocprofv3 kernel trace (csv) by kernel and grid size.
usage: trace_groups.py <t_kernel_trace.csv> [top]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = defaultdict(lambda: [0, 0.0])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    key = (r["Kernel_Name"].split("(")[0][-40:], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))
    g[key][0] += 1
    g[key][1] += d
tot = sum(v[1] for v in g.values())
print(f"total kernel time {tot:.1f} ms in {len(rows)} launches")
for k, v in sorted(g.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k[0]:42s} grid {k[1] // k[2]:9d} x {k[2]:4d}  calls {v[0]:4d}  total {v[1]:9.1f} ms  avg {v[1] / v[0]:9.3f} ms")
