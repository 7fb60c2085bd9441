#!/usr/bin/env python3
"""The Iris classifier's gradient oracles end to end (4 qubits, depth 3, RY/RZ + CZ ring, mini-batches of the 105
training samples): what one optimizer step costs besides the optimizer's own arithmetic, and a short fit.
usage: python tools/bench_oracles_iris.py [calls]"""
import os
import sys
import time

import numpy as np
import torch
from sklearn.datasets import load_iris

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit  # noqa: E402
from vqa_amd.qml_utils import create_ry_rz_cz_ansatz  # noqa: E402
from vqa_amd.qnn_classifier import myQNN_classifier  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 50
data = load_iris()
x = (data.data - data.data.min(0)) / (data.data.max(0) - data.data.min(0)) * np.pi
rng = np.random.default_rng(0)
perm = rng.permutation(len(x))
tr, te = perm[:105], perm[105:]
c = create_ry_rz_cz_ansatz(4, 3, seed=0, circuit_cls=LearningCircuit)
q = myQNN_classifier(4, c, x[tr], data.target[tr], x[te], data.target[te])
p = rng.uniform(0, 2 * np.pi, q.n_params)
for B in (32, 105):
    shots = np.full(q.n_params, 200 * B)
    for label, fn, kw, dev in (("MSE_grad_eval, host statistics", "MSE_grad_eval", {}, False),
                               ("MSE_grad_eval, device statistics", "MSE_grad_eval", {}, True),
                               ("MSE_iEval, device statistics", "MSE_iEval", {}, True)):
        q.device_statistics = dev
        np.random.seed(0)
        getattr(q, fn)(p, shots, mini_batch_size=B, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            getattr(q, fn)(p, shots, mini_batch_size=B, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / calls
        print(f"mini-batch {B:3d}  {label:36s}: {dt * 1e3:8.3f} ms per call", flush=True)
np.random.seed(1)
q.device_statistics = True
q.compile(optimizer="Adam", loss="mean_squared_error", partial_MSE=True,
          hpara=dict(LR_rule=0.1, s_list_rule=np.full(q.n_params, 200), batch_size_rule=32), init_params=p)
t0 = time.perf_counter()
res = q.fit(options=dict(max_iter=100), auto_save=False)
dt = time.perf_counter() - t0
print(f"fit: 100 Adam steps (mini-batch 32, 200 shots per circuit) in {dt:.2f} s = {dt * 10:.2f} ms per step with tracking; "
      f"train loss {res['func_evol'][0]['train loss']:.3f} -> {res['func_evol'][-1]['train loss']:.3f}", flush=True)
