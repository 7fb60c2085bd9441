#!/usr/bin/env python3
"""Where the time of the tiled passes goes: the 20-qubit parameter-shift sweep with parts of
k_tile_pass switched off (results are wrong then; only the time matters).  Needs the
timing-experiment build:  make -C vqa_amd/csrc libvqahip_abl.so
usage: VQA_HIP_LIBRARY=vqa_amd/csrc/libvqahip_abl.so python tools/bench_tile_ablate.py [n depth samples]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit, create_ibm_HEA_ansatz, z_observable  # noqa: E402

n, d, B = (int(v) for v in (sys.argv[1:4] + [20, 8, 2][len(sys.argv) - 1:]))
rng = np.random.default_rng(0)
c = create_ibm_HEA_ansatz(n, d, circuit_cls=LearningCircuit)
x = torch.from_numpy(c.input_angles(rng.uniform(0, np.pi, (B, n)))).cuda()
theta = torch.from_numpy(rng.uniform(0, 2 * np.pi, c.n_theta)).cuda()
obs = [z_observable(n, 0)]
names = {0: "full", 1: "no sweeps (gather + scatter only)", 2: "no LDS traffic in the sweeps", 4: "no register operations",
         8: "no diagonal runs", 12: "LDS traffic only", 6: "neither LDS traffic nor register operations", 10: "register operations only",
         14: "the sweep loop's skeleton only", 16: "no late gates in the tables", 32: "no late gates that join the halves"}
for mode in [int(v) for v in os.environ.get("MODES", "0,1,2,4,8,12,6,10,14,16,32").split(",")]:
    os.environ["VQA_TILE_ABLATE"] = str(mode)
    c.psr_full(theta, x, n_sel=1, obs_list=obs, n_marg_bits=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c.psr_full(theta, x, n_sel=B, obs_list=obs, n_marg_bits=2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / B
    print(f"ablate {mode:2d} {names.get(mode, ''):45s}: {dt * 1e3:7.2f} ms per sample", flush=True)
