#!/bin/bash
# Usage (on the GPU box): tools/prof_vqe.sh <tag> [n depth]: kernel trace of the VQE gradient (one call)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
VQE_REPEAT=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 tools/bench_vqe.py "$@" > $out.log 2>&1
