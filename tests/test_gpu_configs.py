"""The two large benchmark configurations at their real sizes, under -m gpu:

* the 20-qubit QNN circuit (IBM-HEA, depth 8, 900 gates, 540 parameters) with its batch of 1024
  samples: the streaming engine's full sweep (1 107 000 instances) checked through size-independent
  properties, against sweeps of subsets, and against the CPU oracle on seeded instances at depth 8;
* the 28-qubit Pauli-sum circuit (RY + CZ ring, depth 6, 196 parameters, 56-term transverse-field
  Ising ring): exact product-state values, the tiled passes on the whole register (light-cone splitting
  switched off) against the one-sweep-per-gate engine, normalisation, and the parameter-shift gradient
  against central differences.  (A 28-qubit state is
  4 GiB; the CPU oracle is not run at that size.)"""
import numpy as np
import pytest

from helpers import TOL, obs_tuples
from oracle import cpu_oracle
from vqa_amd import LearningCircuit, Observable, _lib
from vqa_amd.qml_utils import create_ibm_HEA_ansatz, create_ry_cz_ansatz
from vqa_amd.vqe_target import PauliSumVQE

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reset_engine():
    yield
    _lib.lib().vqa_set_engine(0)


def z_obs(n, q=0):
    o = Observable(n)
    o.add_operator(1.0, f"Z {q}")
    return o


# ---- 20 qubits, depth 8, batch 1024 -------------------------------------------------------------
def test_twenty_qubit_config_full_batch():
    import torch

    n, depth, B = 20, 8, 1024
    c = create_ibm_HEA_ansatz(n, depth, seed=0)
    assert (c.n_ops, c.n_theta) == (900, 540)
    rng = np.random.default_rng(3)
    theta = rng.uniform(0, 2 * np.pi, c.n_theta)
    x = rng.uniform(0, np.pi, (B, n))
    d_in = c.upload_inputs(x)
    z0 = z_obs(n)
    K = 2 * c.n_theta + 1
    e, m = c.psr_full(theta, d_in, n_sel=B, obs_list=[z0], n_marg_bits=2)
    torch.cuda.synchronize()
    assert e.shape == (B, K, 1) and m.shape == (B, K, 4)
    assert bool(torch.isfinite(e).all()) and bool(torch.isfinite(m).all())
    assert float((m.sum(dim=2) - 1).abs().max()) < TOL
    assert float(m.min()) > -TOL
    assert float((e[:, :, 0] - (m[:, :, 0] - m[:, :, 1] + m[:, :, 2] - m[:, :, 3])).abs().max()) < TOL
    # parameters outside the light cone of the measured qubits: both shifted rows equal the unshifted one
    dead = np.nonzero(c.dead_parameters([z0], 2))[0]
    assert len(dead) > 0
    dcols = torch.from_numpy(np.concatenate([2 * dead + 1, 2 * dead + 2])).cuda()
    assert float((e[:, dcols, 0] - e[:, :1, 0]).abs().max()) == 0.0
    # a sweep of a subset (other chunking of the states in HBM) gives the same rows
    sub = np.sort(rng.choice(B, 3, replace=False)).astype(np.int32)
    e2, m2 = c.psr_full(theta, d_in, sel=sub, obs_list=[z0], n_marg_bits=2)
    rows = torch.from_numpy(sub.astype(np.int64)).cuda()
    assert float((e2 - e[rows]).abs().max()) < TOL and float((m2 - m[rows]).abs().max()) < TOL
    # seeded instances at depth 8 against the CPU oracle: unshifted, live and dead parameters, both signs
    live = np.nonzero(~c.dead_parameters([z0], 2))[0]
    inst = [(int(sub[0]), -1, 0.0)]
    for s in (1, -1):
        for i in rng.choice(live, 3, replace=False):
            inst.append((int(rng.choice(sub)), int(i), s * 0.5 * np.pi))
    inst.append((int(sub[1]), int(dead[0]), 0.5 * np.pi))
    inst.append((int(rng.integers(B)), int(live[-1]), -0.5 * np.pi))
    si = np.array([t[0] for t in inst], np.int32)
    sp = np.array([t[1] for t in inst], np.int32)
    sv = np.array([t[2] for t in inst])
    want_e, want_m = cpu_oracle.eval_batch(c, theta, c.input_angles(x), si, sp, sv, obs_list=obs_tuples([z0]),
                                           n_marg_bits=2, n_threads=9)
    col = np.where(sp < 0, 0, 2 * sp + np.where(sv > 0, 1, 2))
    got_e = e.cpu().numpy()[si, col, 0]
    got_m = m.cpu().numpy()[si, col]
    assert np.abs(got_e - want_e[:, 0]).max() < TOL
    assert np.abs(got_m - want_m).max() < TOL
    # ... and the same instances as an explicit list
    e3, m3 = c.eval_batch(theta, d_in, si, sp, sv, obs_list=[z0], n_marg_bits=2)
    assert np.abs(e3.cpu().numpy() - want_e).max() < TOL and np.abs(m3.cpu().numpy() - want_m).max() < TOL


# ---- 28 qubits, RY + CZ ring, 56-term Ising ring ---------------------------------------------------
def tfim(n, J=1.0, h=1.0):
    H = Observable(n)
    for i in range(n):
        H.add_operator(J, f"Z {i} Z {(i + 1) % n}")
        H.add_operator(h, f"X {i}")
    return H


def test_twenty_eight_qubit_product_state_is_exact():
    """One layer of RY on |0...0>: <Z_i Z_j> = cos(a_i) cos(a_j), <X_i> = -sin(a_i) with
    RY(a) = exp(+i a/2 Y); the energy of the Ising ring is then known in closed form."""
    n = 28
    c = create_ry_cz_ansatz(n, 0, seed=1)
    theta = np.random.default_rng(8).uniform(-np.pi, np.pi, n)
    v = PauliSumVQE(n, c, tfim(n, 0.9, 0.6), min_approx_shots=-10)
    want = sum(0.9 * np.cos(theta[i]) * np.cos(theta[(i + 1) % n]) - 0.6 * np.sin(theta[i]) for i in range(n))
    assert abs(v.exact_energy(theta) - want) < 1e-10


def test_twenty_eight_qubit_config(monkeypatch):
    # (the whole register: the library would otherwise deal the terms into light-cone groups of 13 and 14 qubits,
    # tests/test_gpu_cone_split.py)
    monkeypatch.setenv("VQA_NO_CONE_SPLIT", "1")
    n, depth = 28, 6
    c = create_ry_cz_ansatz(n, depth, seed=4)
    assert c.n_theta == 196
    theta = np.random.default_rng(4).uniform(-np.pi, np.pi, c.n_theta)
    H = tfim(n)
    ident = Observable(n)
    ident.add_operator(1.0, "")
    ia = c.input_angles(np.zeros((1, 1)))
    # the tiled passes against the engine that sweeps the state once per gate: unshifted and two shifted instances
    probe = Observable(n)
    probe.add_operator(0.5, "Z 0 Z 27")
    probe.add_operator(0.25, "Z 13")
    probe.add_operator(-0.7, "X 5")
    probe.add_operator(0.3, "Y 20 Z 21")
    si = np.zeros(3, np.int32)
    sp = np.array([-1, 3, 150], np.int32)
    sv = np.array([0.0, 0.5 * np.pi, -0.5 * np.pi])
    got, gm = c.eval_batch(theta, ia, si, sp, sv, obs_list=[H, probe, ident], n_marg_bits=2)
    assert np.abs(got[:, 2] - 1.0).max() < TOL                  # norm of the final state
    assert np.abs(gm.sum(axis=1) - 1.0).max() < TOL
    _lib.check(_lib.lib().vqa_set_engine(4))
    c4 = create_ry_cz_ansatz(n, depth, seed=4)
    ref, rm = c4.eval_batch(theta, ia, si, sp, sv, obs_list=[H, probe, ident], n_marg_bits=2)
    _lib.check(_lib.lib().vqa_set_engine(0))
    assert np.abs(got - ref).max() < 1e-9 and np.abs(gm - rm).max() < TOL      # 56 terms of order one
    # the full parameter-shift gradient of the energy (392 shifted circuits, 56 terms) against central differences
    v = PauliSumVQE(n, c, H, min_approx_shots=-10)
    e0 = v.exact_energy(theta)
    assert abs(e0 - got[0, 0]) < 1e-9
    g = v.iEval(theta, np.full(c.n_theta, 1000))["grad"]
    assert np.isfinite(g).all()
    for i in (0, 97, 150, c.n_theta - 1):
        d = np.zeros_like(theta)
        d[i] = 1e-4
        fd = (v.exact_energy(theta + d) - v.exact_energy(theta - d)) / 2e-4
        assert abs(g[i] - fd) < 1e-6, (i, g[i], fd)
