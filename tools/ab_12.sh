#!/bin/bash
# usage (GPU box): tools/ab_12.sh tag ...   -- the 12-qubit sweep (and 13) with builds vqa_amd/csrc/libvqahip_x<tag>.so
set -o pipefail
for cfg in "12 10 16384" "13 8 4096"; do
  for t in "$@"; do
    VQA_TAG=$t VQA_HIP_LIBRARY=vqa_amd/csrc/libvqahip_x$t.so timeout -k 10 120 python tools/bench_sweep.py $cfg 3 2>/dev/null | cut -c1-70 || exit 1
  done
done
