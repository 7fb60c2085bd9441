#!/usr/bin/env python3
"""Static instruction mix of one kernel by source line (listing built with -gline-tables-only -S --cuda-device-only).

usage: isa_by_line.py listing.s kernel-name-substring [min-count]
For every source line of the main file (file 1 of the .loc directives... the one named on the command line as
third-last argument is not needed: all files are reported by number), the instructions attributed to it by class:
f64, mfma, dpp, valu (other vector), salu, smem, lds, vmem, scratch."""
import re
import sys
from collections import Counter, defaultdict

path, key = sys.argv[1], sys.argv[2]
minc = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])


def klass(op, line):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"):
        if "dpp" in line or "quad_perm" in line or "row_" in line: return "dpp"
        if op.endswith("_f64") or "_f64_" in op: return "f64"
        return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scr"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
    return "other"


cur = (0, 0)
hist = defaultdict(Counter)
for l in lines[start:end + 1]:
    m = re.match(r"^\s+\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"^\s+([a-z][a-z0-9_]+)", l)
    if not m or l.strip().startswith("."):
        continue
    hist[cur][klass(m.group(1), l)] += 1
cols = ["f64", "mfma", "dpp", "valu", "salu", "smem", "lds", "vmem", "scr", "wait"]
print("file:line   " + " ".join("%5s" % c for c in cols))
for k in sorted(hist):
    h = hist[k]
    if sum(h.values()) >= minc:
        print("%2d:%-6d  " % k + " ".join("%5d" % h[c] for c in cols))
