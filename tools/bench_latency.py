"""Call latency of the host-pointer and device-pointer entry points on the 4-qubit Iris circuit.
usage: python tools/bench_latency.py"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit, z_observable
from vqa_amd.qml_utils import create_ry_rz_cz_ansatz
import torch
c = create_ry_rz_cz_ansatz(4, 3, seed=0)
rng = np.random.default_rng(0)
x = rng.uniform(0, np.pi, (105, 4))
ia = c.input_angles(x)
th = rng.uniform(0, 2*np.pi, c.n_theta)
obs = [z_observable(4, 0)]
for name, f in (("eval_batch host, 1 instance", lambda: c.eval_batch(th, ia[:1], obs_list=obs)),
                ("eval_batch host, 105 instances", lambda: c.eval_batch(th, ia, obs_list=obs)),
                ("psr_full host, 105 samples (6825 evals)", lambda: c.psr_full(th, ia, obs_list=obs, n_marg_bits=2)),
                ):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): f()
    torch.cuda.synchronize()
    print(f"{name:45s} {(time.perf_counter()-t0)/200*1e6:9.1f} us per call")
d_in = c.upload_inputs(x); d_th = torch.from_numpy(th).cuda()
for name, f in (("eval_batch device, 105 instances", lambda: c.eval_batch(d_th, d_in, obs_list=obs)),
                ("psr_full device, 105 samples", lambda: c.psr_full(d_th, d_in, obs_list=obs, n_marg_bits=2))):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): f()
    torch.cuda.synchronize()
    print(f"{name:45s} {(time.perf_counter()-t0)/200*1e6:9.1f} us per call")
