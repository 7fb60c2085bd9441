#!/usr/bin/env python3
"""Plain circuit evaluations per second (no parameter shifts, nothing shared between
instances): the benchmark circuits through vqa_eval_batch_dev, inputs resident.
usage: python tools/bench_evals.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit, create_ibm_HEA_ansatz, z_observable  # noqa: E402

for n, d, B in ((6, 5, 1 << 20), (12, 10, 1 << 18), (20, 8, 256)):
    rng = np.random.default_rng(0)
    c = create_ibm_HEA_ansatz(n, d, circuit_cls=LearningCircuit)
    nf = 1 if n == 6 else n
    x = c.upload_inputs(rng.uniform(0, np.pi, (B, nf)))
    theta = torch.from_numpy(rng.uniform(0, 2 * np.pi, c.n_theta)).cuda()
    obs = [z_observable(n, 0)]
    c.eval_batch(theta, x[:64], obs_list=obs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e, _ = c.eval_batch(theta, x, obs_list=obs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"n={n} depth={d} gates={c.n_ops}: {B / dt:12.0f} circuit evaluations/s ({B} instances, <Z0>), "
          f"unfused state traffic {c.n_ops * 32 * 2 ** n * B / dt / 1e12:.1f} TB/s-equivalent", flush=True)
