#!/usr/bin/env python3
"""profiles/hbm_configs.json (what bench.py prices the 20-qubit configuration with) and a text summary from the
output of tools/prof_hbm_batch.sh.

usage: tools/hbm_configs_profile.py gpurun_out/prof_<tag> <config name> > profiles/<name>.txt

rocprofv3 reports FETCH_SIZE and WRITE_SIZE in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide
(16 B per lane) loads at 64 bytes, so it is doubled (as tools/hbm_traffic.py does for the headline kernel)."""
import collections
import csv
import json
import os
import re
import sys

d, name = sys.argv[1], sys.argv[2]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
args = open(os.path.join(d, "args.txt")).read().split()
n, depth, timed = int(args[0]), int(args[1]), int(args[2])
samples = timed + 1   # the warm-up sample runs under the profiler too
log = open(os.path.join(d, "trace.log")).read()
rate = float(re.search(r"([\d.]+) gradients/s", log).group(1))
passes = float(re.search(r"([\d.]+) state passes per sample", log).group(1))
tr = list(csv.DictReader(open(os.path.join(d, "trace", "trace_kernel_stats.csv"))))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for sub in ("pmc_f64", "pmc_fetch", "pmc_write", "pmc_sq"):
    f = os.path.join(d, sub, "pmc_counter_collection.csv")
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+)", r["Kernel_Name"])
        agg[m.group(1) if m else r["Kernel_Name"][:20]][r["Counter_Name"]] += float(r["Counter_Value"])
tp = agg["k_tile_pass"]
flops = 64.0 * (2.0 * tp["SQ_INSTS_VALU_FMA_F64"] + tp["SQ_INSTS_VALU_MUL_F64"] + tp["SQ_INSTS_VALU_ADD_F64"]) / samples
fetch, write = 2.0 * 1024.0 * tp["FETCH_SIZE"] / samples, 1024.0 * tp["WRITE_SIZE"] / samples
kern_ms = sum(float(r["TotalDurationNs"]) for r in tr if "k_tile_pass" in r["Name"]) / 1e6 / samples
path = os.path.join(root, "profiles", "hbm_configs.json")
try:
    cfg = json.load(open(path))
except (OSError, ValueError):
    cfg = {}
cfg[name] = {
    "fp64_flops_per_sample": flops,
    "state_passes_per_sample": passes,
    "fetch_bytes_per_sample": fetch,
    "write_bytes_per_sample": write,
    "hbm_bytes_per_sample": fetch + write,
    "k_tile_pass_ms_per_sample": kern_ms,
    "valu_wave_instr_per_sample": tp["SQ_INSTS_VALU"] / samples,
    "salu_wave_instr_per_sample": tp["SQ_INSTS_SALU"] / samples,
    "gradients_per_sec_when_profiled": rate,
    "corrections": "FETCH_SIZE x2 (gfx950 wide loads), both counters KiB -> bytes",
    "source": "%s: tools/prof_hbm_batch.sh (tools/bench_hbm_batch.py %d %d %d, %d samples with the warm-up)"
              % (os.path.basename(os.path.normpath(d)), n, depth, timed, samples),
}
json.dump(cfg, open(path, "w"), indent=1)
print("rocprofv3 --kernel-trace --stats and PMC passes (FP64 instruction counters, FETCH_SIZE, WRITE_SIZE, SQ_* shares),")
print("each in a run of its own, of   python3 tools/bench_hbm_batch.py %d %d %d   (tools/prof_hbm_batch.sh)" % (n, depth, timed))
print("IBM-HEA, %d qubits, depth %d, <Z0>: full parameter-shift sweeps of %d samples (one warm-up call of one sample,"
      % (n, depth, samples))
print("one timed call of %d) on the HBM engine.  MI355X.\n" % timed)
for line in log.splitlines():
    if "gradients/s" in line or "state passes" in line:
        print(line)
print("\nkernel                calls   total ms     avg us   share")
for r in tr[:14]:
    m = re.search(r"(k_\w+|__amd_rocclr_\w+)", r["Name"])
    nm = m.group(1) if m else r["Name"][:30]
    tmpl = re.search(r"k_tile_passILi(\d)", r["Name"])
    if tmpl:
        nm += "<%s>" % tmpl.group(1)
    print(f"{nm:24s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.1f} "
          f"{float(r['Percentage']):6.2f}%")
print("\nPMC sums over all dispatches of a run (k_tile_pass = both instantiations):")
for k in ("k_tile_pass",):
    print(f"  {k}: " + ", ".join(f"{a}={b:.4g}" for a, b in sorted(agg[k].items())))
print(f"\nper sample: {flops:.3e} executed FP64 flop, {passes:.1f} state passes, HBM read {fetch / 1e9:.2f} GB + written {write / 1e9:.2f} GB"
      f" (algorithmic: {passes * 2 * 16 * 2 ** n / 1e9:.2f} GB), {kern_ms:.3f} ms in k_tile_pass")
if tp.get("SQ_WAVE_CYCLES"):
    wc = tp["SQ_WAVE_CYCLES"]
    print("shares of SQ_WAVE_CYCLES: " + ", ".join(f"{c[3:]} {100 * tp[c] / wc:.1f} %" for c in
          ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY") if c in tp))
