#!/usr/bin/env python3
"""Full parameter-shift sweeps on the HBM engine in the steady state: the call is repeated at the same batch, so
that the workspace has its final size before the timed call (the first call of a size grows it).
usage: python tools/bench_hbm_steady.py n_qubits depth n_samples [reps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit, create_ibm_HEA_ansatz, z_observable  # noqa: E402

n, d, B = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
rng = np.random.default_rng(3)
c = create_ibm_HEA_ansatz(n, d, circuit_cls=LearningCircuit)
x = torch.from_numpy(c.input_angles(rng.uniform(0, np.pi, (B, n)))).cuda()
theta = torch.from_numpy(rng.uniform(0, 2 * np.pi, c.n_theta)).cuda()
obs = [z_observable(n, 0)]
times = []
for _ in range(reps + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e, _ = c.psr_full(theta, x, n_sel=B, obs_list=obs)
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
best = min(times[1:])
print(f"n={n} d={d} B={B}: first call {B / times[0]:.1f}, then {B / best:.1f} gradients/s ({best / B * 1e3:.3f} ms per sample) "
      f"lanes={os.environ.get('VQA_HBM_GRID_LANES', '-')} streams={os.environ.get('VQA_HBM_STREAMS', '-')}", flush=True)
