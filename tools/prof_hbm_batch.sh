#!/bin/bash
# Usage (on the GPU box): tools/prof_hbm_batch.sh <tag> [n depth samples]
# The 20-qubit QNN configuration as bench.py runs it (tools/bench_hbm_batch.py, <Z0>, one warm-up sample + `samples`
# timed ones): kernel-trace stats, the FP64 instruction counters and the HBM traffic counters, each in a run of its
# own; tools/hbm_configs_profile.py turns the directory into profiles/hbm_configs.json.
set -e
tag=$1; shift
args=${*:-20 8 16}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
echo "$args" > $out/args.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python3 tools/bench_hbm_batch.py $args > $out/trace.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES -d $out/pmc_f64 -o pmc -- python3 tools/bench_hbm_batch.py $args > $out/pmc_f64.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $out/pmc_fetch -o pmc -- python3 tools/bench_hbm_batch.py $args > $out/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $out/pmc_write -o pmc -- python3 tools/bench_hbm_batch.py $args > $out/pmc_write.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $out/pmc_sq -o pmc -- python3 tools/bench_hbm_batch.py $args > $out/pmc_sq.log 2>&1
grep "gradients/s\|state passes" $out/trace.log
