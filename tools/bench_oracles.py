#!/usr/bin/env python3
"""Gradient oracles of the QNN target object end to end (what an optimizer step costs besides
its own arithmetic): MSE_grad_eval and MSE_iEval of the regressor on the sin(x) configuration
(6 qubits, depth 5, 4096 samples per mini-batch), in exact mode, CLT mode with the estimator
algebra on the host (NumPy's generator, the reference's draw order) and CLT mode with the
statistics taken on the GPU.
usage: python tools/bench_oracles.py [n_qubits depth batch calls]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit, create_ibm_HEA_ansatz  # noqa: E402
from vqa_amd.qnn_base import myQNN_base  # noqa: E402

n, d, B, calls = (int(v) for v in (sys.argv[1:5] + [6, 5, 4096, 10][len(sys.argv) - 1:]))
rng = np.random.default_rng(1)
c = create_ibm_HEA_ansatz(n, d, seed=0, circuit_cls=LearningCircuit)
q = myQNN_base(n, c, min_approx_shots=0)
x = rng.uniform(-1, 1, (B, 1))
y = np.sin(np.pi * x[:, 0])
q.set_data(x, y, x[:64], y[:64])
p = np.concatenate([rng.uniform(0, 2 * np.pi, q.n_c_params), [1.0]])
shots = np.full(q.n_params, 1000 * B)
K = 2 * q.n_c_params + 1
print(f"n={n} depth={d} P={q.n_c_params} mini-batch {B}: {B * K} circuit evaluations per gradient", flush=True)
for label, fn, kw, dev in (("MSE_grad_eval exact", "MSE_grad_eval", dict(option_min_approx=-10), False),
                           ("MSE_grad_eval CLT, host statistics", "MSE_grad_eval", {}, False),
                           ("MSE_grad_eval CLT, device statistics", "MSE_grad_eval", {}, True),
                           ("MSE_iEval CLT, host statistics", "MSE_iEval", {}, False),
                           ("MSE_iEval CLT, device statistics", "MSE_iEval", {}, True)):
    q.device_statistics = dev
    np.random.seed(0)
    getattr(q, fn)(p, shots, mini_batch_size=B, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        getattr(q, fn)(p, shots, mini_batch_size=B, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / calls
    print(f"{label:40s}: {dt * 1e3:9.2f} ms per call = {B / dt:12.0f} gradients/s = {B * K / dt:14.0f} evaluations/s", flush=True)
