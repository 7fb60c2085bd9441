#!/usr/bin/env python3
"""The 20-qubit QNN configuration as bench.py runs it (<Z0> alone), a few samples: wall time per
sample beside the kernel time a trace of the same command shows.
usage: python tools/bench_hbm_batch.py [n_qubits depth n_samples]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit, create_ibm_HEA_ansatz, z_observable  # noqa: E402

n, d, B = (int(v) for v in (sys.argv[1:4] + [20, 8, 16][len(sys.argv) - 1:]))
rng = np.random.default_rng(3)
c = create_ibm_HEA_ansatz(n, d, circuit_cls=LearningCircuit)
x = torch.from_numpy(c.input_angles(rng.uniform(0, np.pi, (B, n)))).cuda()
theta = torch.from_numpy(rng.uniform(0, 2 * np.pi, c.n_theta)).cuda()
obs = [z_observable(n, 0)]
c.psr_full(theta, x, n_sel=1, obs_list=obs)
torch.cuda.synchronize()
import ctypes  # noqa: E402
from vqa_amd import _lib  # noqa: E402
rw, ro = ctypes.c_int64(0), ctypes.c_int64(0)
_lib.check(_lib.lib().vqa_state_pass_counters(ctypes.byref(rw), ctypes.byref(ro), 1))
t0 = time.perf_counter()
c.psr_full(theta, x, n_sel=B, obs_list=obs)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
_lib.check(_lib.lib().vqa_state_pass_counters(ctypes.byref(rw), ctypes.byref(ro), 0))
print(f"{(rw.value + ro.value) / B:.1f} state passes per sample ({rw.value / B:.1f} read and write, {ro.value / B:.1f} read only)", flush=True)
print(f"n={n} d={d}: {B} samples in {dt * 1e3:.1f} ms = {dt / B * 1e3:.2f} ms per sample, {B / dt:.1f} gradients/s", flush=True)
