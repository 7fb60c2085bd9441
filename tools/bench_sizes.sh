#!/bin/bash
# Usage (on the GPU box): tools/bench_sizes.sh > gpurun_out/sizes.txt
# Full parameter-shift sweeps of the IBM-HEA circuit with <Z0> at every register size (engine chosen by the library
# unless given), batch sized to a fraction of a second.
set -e
cd $GRAFT_REPO_ROOT
for cfg in "4 3 65536 2" "5 8 65536 2" "6 5 4096 2" "6 5 65536 2" "7 5 65536 2" "8 5 65536 2" "9 6 32768 2" "10 8 16384 2" "11 8 8192 2" \
           "12 10 4096 2" "13 8 2048 2" "14 8 1024 2" "14 8 64 3" "15 8 512 3" "16 8 256 3" "17 8 256 3" "18 8 128 3" "20 8 64 3" "22 8 2 3" "24 8 1 3"; do
    timeout -k 10 120 python tools/bench_engines.py $cfg 2>&1 | grep "gradients/s"
done
