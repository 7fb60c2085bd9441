#!/bin/bash
# HBM-engine sweeps at 15 to 20 qubits, depth 8 (tools/bench_hbm_batch.py); usage: tools/run_sizes_hbm.sh
set -o pipefail
for cfg in "15 512" "16 256" "17 256" "18 128" "20 64"; do
    set -- $cfg
    timeout -k 10 120 python tools/bench_hbm_batch.py $1 8 $2 2>/dev/null || exit 1
done
VQA_TILE12=1 timeout -k 10 120 python tools/bench_hbm_batch.py 20 8 64 2>/dev/null | sed 's/^/12-qubit tiles: /'
