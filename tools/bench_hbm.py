#!/usr/bin/env python3
"""Throughput of the HBM engine on the 20-qubit configuration's circuit (IBM-HEA, depth 8):
tiled passes (engine 3) against one sweep per gate (engine 4).
usage: python tools/bench_hbm.py [n_qubits depth n_evals]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit, _lib, create_ibm_HEA_ansatz, z_observable  # noqa: E402

n, d, E = (int(v) for v in (sys.argv[1:4] or (20, 8, 64)))
for engine, name in ((3, "tiled passes"), (4, "one sweep per gate")):
    rng = np.random.default_rng(0)
    _lib.check(_lib.lib().vqa_set_engine(engine))
    c = create_ibm_HEA_ansatz(n, d, circuit_cls=LearningCircuit)
    x = torch.from_numpy(c.input_angles(rng.uniform(0, np.pi, (E, n)))).cuda()
    theta = torch.from_numpy(rng.uniform(0, 2 * np.pi, c.n_theta)).cuda()
    obs = [z_observable(n, 0)]
    c.eval_batch(theta, x[:2], obs_list=obs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e, _ = c.eval_batch(theta, x, obs_list=obs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    G = c.n_ops
    print(f"{name:20s} n={n} d={d} gates={G}: {E / dt:9.1f} evaluations/s, {dt / E * 1e3:8.3f} ms each; "
          f"unfused state traffic {G * 32 * 2 ** n * E / dt / 1e12:7.2f} TB/s-equivalent; <Z0>[0]={float(e[0, 0]):+.12f}")
_lib.lib().vqa_set_engine(0)
