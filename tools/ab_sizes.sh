#!/bin/bash
# usage (GPU box): tools/ab_sizes.sh tagA tagB ...   -- sweeps at 10..14 qubits with builds vqa_amd/csrc/libvqahip_x<tag>.so
set -o pipefail
for cfg in "10 8 16384" "11 8 8192" "12 10 16384" "13 8 4096" "14 8 1024"; do
  for t in "$@"; do
    VQA_TAG=$t VQA_HIP_LIBRARY=vqa_amd/csrc/libvqahip_x$t.so timeout -k 10 120 python tools/bench_sweep.py $cfg 3 2>/dev/null | cut -c1-70 || exit 1
  done
done
