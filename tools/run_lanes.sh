#!/bin/bash
# 15- to 20-qubit sweeps with different numbers of samples per launch and streams (tools/bench_hbm_batch.py)
# usage: tools/run_lanes.sh   (on a GPU box, from the repository root)
set -o pipefail
run() {  # n batch lanes streams
    VQA_HBM_GRID_LANES=$3 VQA_HBM_STREAMS=$4 timeout -k 10 120 python tools/bench_hbm_batch.py $1 8 $2 2>/dev/null | sed "s/^/lanes=$3 streams=$4: /" || return 1
}
for cfg in "15 512" "16 256" "17 256"; do
    set -- $cfg
    for ls in "1 4" "4 1" "8 1" "16 1" "8 2" "16 2" "16 4" "32 2"; do
        set -- $cfg $ls
        run $1 $2 $3 $4 || exit 1
    done
done
for cfg in "18 128" "20 64"; do
    for ls in "1 4" "4 1" "8 1" "4 2" "8 2" "4 4"; do
        set -- $cfg $ls
        run $1 $2 $3 $4 || exit 1
    done
done
