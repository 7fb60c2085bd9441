#!/usr/bin/env python3
"""Light-cone splitting at the 28-qubit configuration: the sweep of all 56 Ising terms through the device-pointer
entry point, time per call (one call = one full parameter-shift gradient's [2P+1, 56] matrix).
usage: python tools/bench_split.py [n_qubits depth reps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit, Observable  # noqa: E402
from vqa_amd.qml_utils import create_ry_cz_ansatz  # noqa: E402

n, d, reps = (int(v) for v in (sys.argv[1:4] + [28, 6, 20][len(sys.argv) - 1:]))
c = create_ry_cz_ansatz(n, d, seed=4, circuit_cls=LearningCircuit)
obs = []
for i in range(n):
    for s in (f"Z {i} Z {(i + 1) % n}", f"X {i}"):
        o = Observable(n)
        o.add_operator(1.0, s)
        obs.append(o)
theta = torch.from_numpy(np.random.default_rng(4).uniform(-np.pi, np.pi, c.n_theta)).cuda()
x = c.upload_inputs(np.zeros((1, 1)))
t0 = time.perf_counter()
e, _ = c.psr_full(theta, x, n_sel=1, obs_list=obs)
torch.cuda.synchronize()
print(f"first call (plans, programs): {time.perf_counter() - t0:.3f} s", flush=True)
for _ in range(3):
    c.psr_full(theta, x, n_sel=1, obs_list=obs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    c.psr_full(theta, x, n_sel=1, obs_list=obs, out_expect=e)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"n={n} depth={d} P={c.n_theta}: {1e3 * (t2 - t0) / reps:.3f} ms per sweep ({1e3 * (t1 - t0) / reps:.3f} ms of it queuing on the host)")
