#!/usr/bin/env python3
"""Full parameter-shift sweeps of one circuit size on each engine that takes it.
usage: python tools/bench_engines.py n_qubits depth n_samples [engines...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit, _lib, create_ibm_HEA_ansatz, z_observable  # noqa: E402

n, d, B = (int(v) for v in sys.argv[1:4])
for eng in [int(v) for v in sys.argv[4:]] or [1, 2, 3]:
    try:
        _lib.check(_lib.lib().vqa_set_engine(eng))
        c = create_ibm_HEA_ansatz(n, d, circuit_cls=LearningCircuit)
        rng = np.random.default_rng(0)
        x = torch.from_numpy(c.input_angles(rng.uniform(0, np.pi, (B, n)))).cuda()
        theta = torch.from_numpy(rng.uniform(0, 2 * np.pi, c.n_theta)).cuda()
        obs = [z_observable(n, 0)]
        # (the same call once before the timed one: the first call of a size builds plans and grows the workspace)
        c.psr_full(theta, x, n_sel=B, obs_list=obs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e, _ = c.psr_full(theta, x, n_sel=B, obs_list=obs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"n={n} d={d} engine {eng}: {B / dt:10.1f} gradients/s ({B * (2 * c.n_theta + 1) / dt:12.0f} evaluations/s), "
              f"<Z0>[0,0]={float(e[0, 0, 0]):+.10f}", flush=True)
        c.eval_batch(theta, x[:2], obs_list=obs)
        torch.cuda.synchronize()
        reps = max(1, min(256, (1 << 20) // B))
        t0 = time.perf_counter()
        for _ in range(reps):
            c.eval_batch(theta, x, obs_list=obs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"n={n} d={d} engine {eng}: {reps * B / dt:12.0f} plain evaluations/s (explicit list of {B}, {reps} calls)", flush=True)
    except _lib.VqaError as err:
        print(f"n={n} d={d} engine {eng}: {err}", flush=True)
_lib.lib().vqa_set_engine(0)
