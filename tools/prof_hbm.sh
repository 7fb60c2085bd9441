#!/bin/bash
# Usage (on the GPU box): tools/prof_hbm.sh <tag>
# kernel-trace stats and HBM traffic counters of a 20-qubit parameter-shift sweep on the HBM engine
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python3 tools/bench_hbm_sweep.py 20 8 1 64 > $out/trace.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $out/pmc_fetch -o pmc -- python3 tools/bench_hbm_sweep.py 20 8 1 64 > $out/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $out/pmc_write -o pmc -- python3 tools/bench_hbm_sweep.py 20 8 1 64 > $out/pmc_write.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $out/pmc_sq -o pmc -- python3 tools/bench_hbm_sweep.py 20 8 1 64 > $out/pmc_sq.log 2>&1
find $out -name "*.csv" | head -20
cat $out/trace.log | tail -4
