#!/usr/bin/env python3
"""Text summary of tools/prof_hbm.sh output (kernel-trace stats + PMC passes of a 20-qubit sweep).
usage: tools/hbm_profile_summary.py gpurun_out/prof_<tag> > profiles/<name>.txt"""
import collections
import csv
import os
import re
import sys

d = sys.argv[1]
print("rocprofv3 --kernel-trace --stats (+ PMC passes FETCH_SIZE / WRITE_SIZE / SQ_*) of")
print("  python3 tools/bench_hbm_sweep.py 20 8 1 64     (tools/prof_hbm.sh)")
print("IBM-HEA, 20 qubits, depth 8 (900 gates, P = 540), <Z0> and the two-bit marginal: two full parameter-shift")
print("sweeps of one sample (one warm-up, one timed) on the HBM engine + 64 explicit instances.  MI355X.\n")
log = os.path.join(d, "trace.log")
for line in open(log):
    if line.startswith(("sweep", "explicit", "max |")):
        print(line.rstrip())
print("\nkernel                calls   total ms     avg us   share")
tr = list(csv.DictReader(open(os.path.join(d, "trace", "trace_kernel_stats.csv"))))
for r in tr[:12]:
    m = re.search(r"(k_\w+|__amd_rocclr_\w+)", r["Name"])
    nm = m.group(1) if m else r["Name"][:30]
    print(f"{nm:24s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.1f} "
          f"{float(r['Percentage']):6.2f}%")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for name in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for r in csv.DictReader(open(os.path.join(d, name, "pmc_counter_collection.csv"))):
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        agg[m.group(1) if m else r["Kernel_Name"][:20]][r["Counter_Name"]] += float(r["Counter_Value"])
print("\nPMC sums over all dispatches of the run (separate passes; k_tile_pass = both instantiations, with and")
print("without the measurement at the end of the pass):")
for k in ("k_tile_pass", "k_zpart", "k_prep_tab"):
    if k in agg:
        print(f"  {k}: " + ", ".join(f"{a}={b:.4g}" for a, b in sorted(agg[k].items())))
tp = agg["k_tile_pass"]
fetch, write = tp["FETCH_SIZE"] * 1024 * 2, tp["WRITE_SIZE"] * 1024
tms = sum(float(r["TotalDurationNs"]) for r in tr if "k_tile_pass" in r["Name"]) / 1e9
print(f"\nk_tile_pass HBM traffic (FETCH_SIZE x2 for gfx950's wide loads, KiB -> bytes): read {fetch / 1e9:.1f} GB + "
      f"written {write / 1e9:.1f} GB")
print(f"  in {tms * 1e3:.1f} ms of kernel time = {(fetch + write) / tms / 1e12:.2f} TB/s (8 TB/s peak; 6.0-6.2 TB/s copy rate "
      "measured on the box)")
print(f"  LDS bank-conflict cycles / LDS-active cycles = {tp['SQ_LDS_BANK_CONFLICT'] / tp['SQ_ACTIVE_INST_LDS']:.3f}")
valu_cyc = tp["SQ_INSTS_VALU"] * 4 / 1024 / 2.4e9
print(f"  vector instructions: {tp['SQ_INSTS_VALU']:.4g} of a wave = {valu_cyc * 1e3:.1f} ms of issue at 4 cycles each on 1024 SIMDs at 2.4 GHz "
      f"({valu_cyc / tms:.0%} of the kernel time)")
