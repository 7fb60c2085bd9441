import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from vqa_amd import LearningCircuit, Observable
from vqa_amd.qml_utils import create_ry_cz_ansatz
from vqa_amd.vqe_target import PauliSumVQE
n, d = 28, 6
c = create_ry_cz_ansatz(n, d, seed=4, circuit_cls=LearningCircuit)
h = Observable(n)
for i in range(n):
    h.add_operator(1.0, f"Z {i} Z {(i + 1) % n}")
    h.add_operator(1.0, f"X {i}")
vqe = PauliSumVQE(n, c, h, min_approx_shots=-10)
theta = np.random.default_rng(4).uniform(-np.pi, np.pi, c.n_theta)
if os.environ.get("ENERGY_FIRST"):
    vqe.exact_energy(theta)
for k in range(int(os.environ.get("CALLS", "3"))):
    f, t = torch.cuda.mem_get_info()
    t0 = time.perf_counter()
    out = vqe.iEval(theta, np.full(c.n_theta, 1000))
    torch.cuda.synchronize()
    print(f"call {k}: {time.perf_counter() - t0:.3f} s; free before {f / 2**30:.1f} GiB of {t / 2**30:.1f}", flush=True)
