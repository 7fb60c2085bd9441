#!/bin/bash
# Usage (on the GPU box): tools/prof.sh <tag> <bench args...>
# kernel-trace stats + two PMC passes for the bench command; summaries land in gpurun_out/prof_<tag>/
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- python3 bench.py "$@" --no-cpu-baseline --no-other-configs > $out/trace.log 2>&1
timeout -s KILL 300 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY -d $out/pmc1 -o pmc1 -- python3 bench.py "$@" --no-cpu-baseline --no-other-configs > $out/pmc1.log 2>&1
timeout -s KILL 300 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out/pmc2 -o pmc2 -- python3 bench.py "$@" --no-cpu-baseline --no-other-configs > $out/pmc2.log 2>&1
find $out -name "*.csv" | head -20
timeout -s KILL 300 rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_LEVEL_WAVES -d $out/pmc3 -o pmc3 -- python3 bench.py "$@" --no-cpu-baseline --no-other-configs > $out/pmc3.log 2>&1
# HBM traffic: FETCH_SIZE and WRITE_SIZE do not fit one pass
timeout -s KILL 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $out/pmc4 -o pmc4 -- python3 bench.py "$@" --no-cpu-baseline --no-other-configs > $out/pmc4.log 2>&1
timeout -s KILL 300 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $out/pmc5 -o pmc5 -- python3 bench.py "$@" --no-cpu-baseline --no-other-configs > $out/pmc5.log 2>&1
# matrix-unit FP64 (the lane rotations on bits 4 and 5)
timeout -s KILL 300 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $out/pmc6 -o pmc6 -- python3 bench.py "$@" --no-cpu-baseline --no-other-configs > $out/pmc6.log 2>&1
