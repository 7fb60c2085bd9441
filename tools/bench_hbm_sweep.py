#!/usr/bin/env python3
"""Full parameter-shift sweeps on the HBM engine: the sweep path (one moved state per
parameter, passes in front of its rotation shared) against the explicit instance list.
usage: python tools/bench_hbm_sweep.py [n_qubits depth n_samples explicit_instances]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit, create_ibm_HEA_ansatz, z_observable  # noqa: E402
from oracle import cpu_oracle  # noqa: E402  (instance lists only)

n, d, B, EX = (int(v) for v in (sys.argv[1:5] + [20, 8, 2, 128][len(sys.argv) - 1:]))
rng = np.random.default_rng(0)
c = create_ibm_HEA_ansatz(n, d, circuit_cls=LearningCircuit)
P = c.n_theta
K = 2 * P + 1
x = torch.from_numpy(c.input_angles(rng.uniform(0, np.pi, (B, n)))).cuda()
theta = torch.from_numpy(rng.uniform(0, 2 * np.pi, P)).cuda()
obs = [z_observable(n, 0)]
c.psr_full(theta, x, n_sel=1, obs_list=obs, n_marg_bits=2)
torch.cuda.synchronize()
t0 = time.perf_counter()
e, m = c.psr_full(theta, x, n_sel=B, obs_list=obs, n_marg_bits=2)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"sweep         n={n} d={d} P={P}: {B / dt:8.3f} gradients/s = {B * K / dt:10.1f} evaluations/s "
      f"({dt / B * 1e3:.1f} ms per sample)", flush=True)
# a slice of the same instances through the explicit list
si, sp, sv = cpu_oracle.psr_lists(B, P)
pick = np.sort(rng.choice(B * K, min(EX, B * K), replace=False))
c.eval_batch(theta, x, si[pick[:2]], sp[pick[:2]], sv[pick[:2]], obs_list=obs, n_marg_bits=2)
torch.cuda.synchronize()
t0 = time.perf_counter()
e2, m2 = c.eval_batch(theta, x, si[pick], sp[pick], sv[pick], obs_list=obs, n_marg_bits=2)
torch.cuda.synchronize()
dt2 = time.perf_counter() - t0
print(f"explicit list n={n} d={d}: {len(pick) / dt2:10.1f} evaluations/s = {len(pick) / dt2 / K:8.3f} gradients/s", flush=True)
de = (e.reshape(B * K, -1)[pick] - e2).abs().max().item()
dm = (m.reshape(B * K, -1)[pick] - m2).abs().max().item()
print(f"max |sweep - explicit| over {len(pick)} instances: expectation {de:.2e}, marginal {dm:.2e}", flush=True)
