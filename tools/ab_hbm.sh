#!/bin/bash
# usage (GPU box): tools/ab_hbm.sh tag ...   -- steady-state sweeps on the HBM engine with builds vqa_amd/csrc/libvqahip_x<tag>.so
set -o pipefail
for cfg in "15 8 512" "16 8 256" "20 8 64"; do
  for t in "$@"; do
    echo -n "$t: "; VQA_HIP_LIBRARY=vqa_amd/csrc/libvqahip_x$t.so timeout -k 10 120 python tools/bench_hbm_steady.py $cfg 2>/dev/null | cut -c1-100 || exit 1
  done
done
