#!/bin/bash
# Usage (on the GPU box): [PMC="counters ..."] tools/pmc_sweep.sh <tag> <library or ""> <n> <depth> <batch>
# One counter pass over tools/bench_sweep.py (2 calls of the sweep); per-kernel sums land in gpurun_out/pmc_<tag>.txt
tag=$1; lib=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
[ -n "$lib" ] && export VQA_HIP_LIBRARY=$lib
export VQA_TAG=$tag
timeout -s KILL 120 rocprofv3 --output-format csv --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY} -d $out -o p -- python3 tools/bench_sweep.py "$@" 1 > $out/log.txt 2>&1 || exit 1
python3 tools/pmc_summary.py $out > gpurun_out/pmc_$tag.txt 2>&1 || python3 - $out > gpurun_out/pmc_$tag.txt <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in tot.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v:.4g}")
PY
grep -A12 k_reg_psr gpurun_out/pmc_$tag.txt | head -14
