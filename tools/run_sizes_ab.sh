# usage: tools/run_sizes_ab.sh out.txt ENVVAR   -- sweeps at 10..13 qubits with and without ENVVAR=1
O=gpurun_out/$1; rm -f $O
for cfg in "10 8 16384" "11 8 8192" "12 10 16384" "13 8 2048"; do
  VQA_TAG=on timeout -k 10 120 python tools/bench_sweep.py $cfg 3 >> $O 2>&1 || exit 1
  env $2=1 VQA_TAG=$2 timeout -k 10 120 python tools/bench_sweep.py $cfg 3 >> $O 2>&1 || exit 1
done
grep gradients $O | cut -c1-90
