"""Not collected by pytest: a longer run of tests/test_gpu_parity.py::test_streaming_sweep_random_circuits.
Full parameter-shift sweeps of 60 random circuits (13 to 17 qubits) on the HBM engine, in both tile
shapes, whole and in chunks of parameters, with and without X / Y terms, every row against the oracle.
usage (on a GPU box, from the repository root): python tests/fuzz_hbm_sweeps.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from vqa_amd import LearningCircuit, _lib
from oracle import cpu_oracle
from helpers import random_sweep_circuit, random_observables, obs_tuples
_lib.check(_lib.lib().vqa_set_engine(3))
bad = 0
for seed in range(100, 160):
    n = 13 + seed % 5
    for tile12 in (False, True):
        for cap in (None, 14):
            if tile12: os.environ["VQA_TILE12"] = "1"
            else: os.environ.pop("VQA_TILE12", None)
            if cap: os.environ["VQA_HBM_MAX_STATES"] = str(cap)
            else: os.environ.pop("VQA_HBM_MAX_STATES", None)
            c = random_sweep_circuit(LearningCircuit, n, 60 + seed % 50, seed=seed, n_theta=10 + seed % 20)
            rng = np.random.default_rng(seed)
            theta = rng.uniform(-3, 3, c.n_theta); x = rng.uniform(-1, 1, (2, 3)); ia = c.input_angles(x)
            obs = random_observables(n, seed=seed, n_obs=2, max_terms=3, xy=(seed % 3 == 0))
            si, sp, sv = cpu_oracle.psr_lists(2, c.n_theta)
            we, wm = cpu_oracle.eval_batch(c, theta, ia, si, sp, sv, obs_list=obs_tuples(obs), n_marg_bits=2)
            ge, gm = c.psr_full(theta, ia, obs_list=obs, n_marg_bits=2)
            err = max(np.abs(np.asarray(ge).reshape(we.shape) - we).max(), np.abs(np.asarray(gm).reshape(wm.shape) - wm).max())
            if not err < 1e-10:
                bad += 1
                print("MISMATCH", seed, n, tile12, cap, err, flush=True)
print("fuzz done, mismatches:", bad)
