#!/usr/bin/env python3
"""The VQE-shaped configuration: RY + CZ-ring ansatz (depth 6), transverse-field Ising
Hamiltonian on a ring (2n Pauli terms), full parameter-shift gradient of the energy through
the Pauli-sum target object.
usage: python tools/bench_vqe.py [n_qubits depth]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit, Observable  # noqa: E402
from vqa_amd.qml_utils import create_ry_cz_ansatz  # noqa: E402
from vqa_amd.vqe_target import PauliSumVQE  # noqa: E402

n, d = (int(v) for v in (sys.argv[1:3] + [28, 6][len(sys.argv) - 1:]))
c = create_ry_cz_ansatz(n, d, seed=4, circuit_cls=LearningCircuit)
h = Observable(n)
for i in range(n):
    h.add_operator(1.0, f"Z {i} Z {(i + 1) % n}")
    h.add_operator(1.0, f"X {i}")
vqe = PauliSumVQE(n, c, h, min_approx_shots=-10)
theta = np.random.default_rng(4).uniform(-np.pi, np.pi, c.n_theta)
t0 = time.perf_counter()
e0 = vqe.exact_energy(theta)
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"n={n} depth={d} P={c.n_theta} terms={len(vqe.terms)}: energy {e0:+.10f} in {t1 - t0:.2f} s (first call)", flush=True)
if n <= 20:   # small registers: leave the one-time plans and programs of the sweep out of the timing
    vqe.iEval(theta, np.full(c.n_theta, 1000))
    torch.cuda.synchronize()
t0 = time.perf_counter()
out = vqe.iEval(theta, np.full(c.n_theta, 1000))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if n > 20 and os.environ.get("VQE_REPEAT", "1") != "0":   # once more: the first call grows the library's workspace
    print(f"  (first call: {dt:.4f} s)", flush=True)
    t0 = time.perf_counter()
    out = vqe.iEval(theta, np.full(c.n_theta, 1000))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
g = out["grad"]
print(f"full parameter-shift gradient ({2 * c.n_theta} shifted evaluations x {len(vqe.terms)} terms): {dt:.4f} s "
      f"= {2 * c.n_theta / dt:.1f} evaluations/s; |g| = {np.linalg.norm(g):.8f}", flush=True)
# finite-difference check of two components against the exact energy
for i in (0, c.n_theta // 2):
    e = np.zeros_like(theta); e[i] = 1e-4
    fd = (vqe.exact_energy(theta + e) - vqe.exact_energy(theta - e)) / 2e-4
    print(f"  component {i}: parameter shift {g[i]:+.8f}, central difference {fd:+.8f}", flush=True)
