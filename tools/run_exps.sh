O=gpurun_out/$1; shift; rm -f $O
for e in "$@"; do
  VQA_TAG=x$e VQA_HIP_LIBRARY=vqa_amd/csrc/libvqahip_x$e.so timeout -k 10 120 python tools/bench_sweep.py 12 10 16384 3 >> $O 2>&1 || exit 1
done
grep gradients $O | cut -c1-80
