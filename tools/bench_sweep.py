#!/usr/bin/env python3
"""Times full parameter-shift sweeps of the IBM-HEA circuit with <Z0> (best of `reps` calls) and prints a
checksum of the outputs, for comparing builds of the library (VQA_HIP_LIBRARY) and run-time switches.
usage: python tools/bench_sweep.py n_qubits depth n_samples [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit, create_ibm_HEA_ansatz, z_observable  # noqa: E402

n, d, B = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
c = create_ibm_HEA_ansatz(n, d, circuit_cls=LearningCircuit)
rng = np.random.default_rng(0)
x = torch.from_numpy(c.input_angles(rng.uniform(0, np.pi, (B, n)))).cuda()
theta = torch.from_numpy(rng.uniform(0, 2 * np.pi, c.n_theta)).cuda()
obs = [z_observable(n, 0)]
c.psr_full(theta, x, n_sel=min(B, 2), obs_list=obs)
torch.cuda.synchronize()
best = 1e30
for _ in range(reps):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    e, _ = c.psr_full(theta, x, n_sel=B, obs_list=obs)
    ev1.record()
    torch.cuda.synchronize()
    best = min(best, ev0.elapsed_time(ev1) * 1e-3)
tag = os.environ.get("VQA_TAG", "")
print(f"{tag} n={n} d={d} B={B}: {B / best:10.1f} gradients/s  ({best * 1e3:8.2f} ms)  checksum {float(e.double().sum()):+.12f} "
      f"abs {float(e.abs().sum()):.12f}", flush=True)
