#!/usr/bin/env python3
"""Sweeps with an observable that holds X terms (transverse-field Ising ring), by batch size: the
register engine behind basis-change gates against the direct X/Y measurement of the LDS /
streaming engines.
usage: python tools/bench_xy.py n_qubits depth"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit, Observable, create_ibm_HEA_ansatz  # noqa: E402

n, d = int(sys.argv[1]), int(sys.argv[2])
c = create_ibm_HEA_ansatz(n, d, circuit_cls=LearningCircuit)
h = Observable(n)
for i in range(n):
    h.add_operator(1.0, f"Z {i} Z {(i + 1) % n}")
    h.add_operator(1.0, f"X {i}")
rng = np.random.default_rng(0)
theta = torch.from_numpy(rng.uniform(0, 2 * np.pi, c.n_theta)).cuda()
for B in (1, 4, 16, 64, 256, 1024):
    x = torch.from_numpy(c.input_angles(rng.uniform(0, np.pi, (B, n)))).cuda()
    res = []
    for env in ("", "1"):
        if env:
            os.environ["VQA_NO_BASIS_CHANGE"] = env
        else:
            os.environ.pop("VQA_NO_BASIS_CHANGE", None)
        c.psr_full(theta, x, obs_list=[h])
        torch.cuda.synchronize()
        reps = 3 if B >= 256 else 10
        t0 = time.perf_counter()
        for _ in range(reps):
            e, _ = c.psr_full(theta, x, obs_list=[h])
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / reps)
    os.environ.pop("VQA_NO_BASIS_CHANGE", None)
    K = 2 * c.n_theta + 1
    print(f"n={n} d={d} B={B:5d} ({B * K:8d} instances): basis change {res[0] * 1e3:9.3f} ms, direct {res[1] * 1e3:9.3f} ms", flush=True)
