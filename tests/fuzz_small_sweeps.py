"""Not collected by pytest: full parameter-shift sweeps of random circuits on 4 to 14 qubits with the
engine the library chooses (register engine: grouped sweeps below 10 qubits, the sweep kernel above; LDS
engine when the call has X / Y terms), a few batch sizes, every row against the oracle.
usage (on a GPU box, from the repository root): python tests/fuzz_small_sweeps.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit, _lib  # noqa: E402
from oracle import cpu_oracle  # noqa: E402
from helpers import random_sweep_circuit, random_observables, obs_tuples  # noqa: E402

_lib.check(_lib.lib().vqa_set_engine(0))
bad = 0
for seed in range(200, 310):
    n = 4 + seed % 11
    B = (1, 3, 17)[seed % 3]
    c = random_sweep_circuit(LearningCircuit, n, 30 + seed % 60, seed=seed, n_theta=6 + seed % 20)
    rng = np.random.default_rng(seed)
    theta = rng.uniform(-3, 3, c.n_theta)
    x = rng.uniform(-1, 1, (B, 3))
    ia = c.input_angles(x)
    obs = random_observables(n, seed=seed, n_obs=2, max_terms=3, xy=(seed % 7 == 0))
    si, sp, sv = cpu_oracle.psr_lists(B, c.n_theta)
    we, wm = cpu_oracle.eval_batch(c, theta, ia, si, sp, sv, obs_list=obs_tuples(obs), n_marg_bits=min(n, 2))
    ge, gm = c.psr_full(theta, ia, obs_list=obs, n_marg_bits=min(n, 2))
    err = max(np.abs(np.asarray(ge).reshape(we.shape) - we).max(), np.abs(np.asarray(gm).reshape(wm.shape) - wm).max())
    if not err < 1e-10:
        bad += 1
        print("MISMATCH seed", seed, "n", n, "B", B, "engine", c.engine if hasattr(c, "engine") else "?", "err", err, flush=True)
print("fuzz done, mismatches:", bad)
