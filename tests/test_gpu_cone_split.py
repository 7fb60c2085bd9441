"""Light-cone splitting of a call's Pauli terms (vqa_abi.hip, split_plan / split_call): a sum of local terms
behind a shallow circuit on a wide register is evaluated group by group on circuits reduced to the groups'
backward light cones.  Checked against the CPU oracle on the whole register, against the library with the
splitting switched off, and at 28 qubits against the engine that sweeps the whole state once per gate."""
import os
from contextlib import contextmanager

import numpy as np
import pytest

from helpers import TOL, obs_tuples
from oracle import cpu_oracle
from vqa_amd import LearningCircuit, Observable, _lib
from vqa_amd.qml_utils import create_ibm_HEA_ansatz, create_ry_cz_ansatz

pytestmark = pytest.mark.gpu


@contextmanager
def env(**kw):
    old = {k: os.environ.get(k) for k in kw}
    os.environ.update({k: str(v) for k, v in kw.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def local_terms(n, with_y=True):
    """One observable per term: ZZ ring, X on every qubit, a few Y / mixed strings."""
    out = []
    for i in range(n):
        o = Observable(n)
        o.add_operator(1.0, f"Z {i} Z {(i + 1) % n}")
        out.append(o)
        o = Observable(n)
        o.add_operator(0.7, f"X {i}")
        out.append(o)
    if with_y:
        for s in (f"Y 2", f"Y {n - 1} Z 0", f"X 3 Y 4", f"Z 5 X 6 Z 7"):
            o = Observable(n)
            o.add_operator(-0.4, s)
            out.append(o)
    return out


def sweep_instances(P, j=0):
    si = np.full(2 * P + 1, j, np.int32)
    sp = np.concatenate([[-1], np.repeat(np.arange(P), 2)]).astype(np.int32)
    sv = np.concatenate([[0.0], np.tile([0.5 * np.pi, -0.5 * np.pi], P)])
    return si, sp, sv


def oracle_sweep(c, theta, x, obs, n_marg_bits=0, j=0):
    si, sp, sv = sweep_instances(c.n_theta, j)
    return cpu_oracle.eval_batch(c, theta, c.input_angles(x), si, sp, sv, obs_list=obs_tuples(obs),
                                 n_marg_bits=n_marg_bits, n_threads=8)


@pytest.mark.parametrize("n,depth", [(16, 2), (17, 1)])
def test_split_sweep_against_oracle(n, depth):
    c = create_ry_cz_ansatz(n, depth, seed=3)
    theta = np.random.default_rng(n).uniform(-np.pi, np.pi, c.n_theta)
    x = np.zeros((1, 1))
    obs = local_terms(n)
    want, _ = oracle_sweep(c, theta, x, obs)
    got, _ = c.psr_full(theta, c.input_angles(x), n_sel=1, obs_list=obs)
    assert got.shape == (1, 2 * c.n_theta + 1, len(obs))
    assert np.abs(got[0] - want).max() < TOL
    # the same call with the splitting switched off: the streaming engine on the whole register
    with env(VQA_NO_CONE_SPLIT=1):
        c2 = create_ry_cz_ansatz(n, depth, seed=3)
        whole, _ = c2.psr_full(theta, c2.input_angles(x), n_sel=1, obs_list=obs)
    assert np.abs(got - whole).max() < TOL


def test_split_one_observable_of_many_terms_and_marginal():
    """All terms in ONE observable (the groups' parts are added up), an identity term, and the marginal of the two
    low qubits, which one of the groups measures."""
    n, depth = 16, 2
    c = create_ry_cz_ansatz(n, depth, seed=5)
    theta = np.random.default_rng(1).uniform(-np.pi, np.pi, c.n_theta)
    x = np.zeros((1, 1))
    H = Observable(n)
    for i in range(n):
        H.add_operator(0.9, f"Z {i} Z {(i + 1) % n}")
        H.add_operator(-0.6, f"X {i}")
    H.add_operator(0.25, "")
    H.add_operator(0.3, "Y 9 Y 10")
    one = Observable(n)
    one.add_operator(2.0, "Z 12")
    want_e, want_m = oracle_sweep(c, theta, x, [H, one], n_marg_bits=2)
    got_e, got_m = c.psr_full(theta, c.input_angles(x), n_sel=1, obs_list=[H, one], n_marg_bits=2)
    assert np.abs(got_e[0] - want_e).max() < 10 * TOL      # 34 terms of order one
    assert np.abs(got_m[0] - want_m).max() < TOL
    # explicit list
    si, sp, sv = sweep_instances(c.n_theta)
    pick = np.array([0, 1, 2, 17, 40, 2 * c.n_theta])
    e2, m2 = c.eval_batch(theta, c.input_angles(x), si[pick], sp[pick], sv[pick], obs_list=[H, one], n_marg_bits=2)
    assert np.abs(e2 - want_e[pick]).max() < 10 * TOL and np.abs(m2 - want_m[pick]).max() < TOL


def test_split_with_inputs_and_samples():
    """A data-dependent circuit on a wide register with a shallow trainable part: several samples, selected rows."""
    n = 15
    c = LearningCircuit(n)
    for q in range(n):
        c.add_input_RY_gate(q, lambda x, q=q: x[q % len(x)] * (1 + 0.05 * q))
    for layer in range(2):
        for q in range(n):
            c.add_parametric_RX_gate(q, 0.1 * q + layer)
            c.add_parametric_RZ_gate(q, 0.2 * q - layer)
        for q in range(0, n - 1, 2):
            c.add_CNOT_gate(q, q + 1)
        for q in range(1, n - 1, 2):
            c.add_CZ_gate(q, q + 1)
    rng = np.random.default_rng(2)
    theta = rng.uniform(-np.pi, np.pi, c.n_theta)
    x = rng.uniform(0, np.pi, (5, 3))
    obs = []
    for q in (0, 7, 14):
        o = Observable(n)
        o.add_operator(1.0, f"Z {q}")
        obs.append(o)
    o = Observable(n)
    o.add_operator(0.5, "X 3 X 4")
    o.add_operator(0.5, "Z 10")
    obs.append(o)
    sel = np.array([4, 1, 3], np.int32)
    got, gm = c.psr_full(theta, c.input_angles(x), sel=sel, obs_list=obs, n_marg_bits=1)
    for r, j in enumerate(sel):
        want, wm = oracle_sweep(c, theta, x, obs, n_marg_bits=1, j=int(j))
        assert np.abs(got[r] - want).max() < TOL
        assert np.abs(gm[r] - wm).max() < TOL


def test_split_parameter_subset():
    n, depth = 16, 2
    c = create_ry_cz_ansatz(n, depth, seed=7)
    theta = np.random.default_rng(7).uniform(-np.pi, np.pi, c.n_theta)
    ia = c.input_angles(np.zeros((1, 1)))
    obs = local_terms(n, with_y=False)
    full, _ = c.psr_full(theta, ia, n_sel=1, obs_list=obs)
    mine = np.arange(1, c.n_theta, 3)
    part, _ = c.psr_full(theta, ia, n_sel=1, obs_list=obs, params=mine)
    rows = np.concatenate([[0], 2 * mine + 1, 2 * mine + 2])
    assert np.abs(part[0, rows] - full[0, rows]).max() < TOL


def test_split_ibm_hea_wide_and_shallow():
    """The reference's IBM-HEA ansatz at depth 0 on 15 qubits: <Z7> behind the feature map."""
    n = 15
    c = create_ibm_HEA_ansatz(n, 0, seed=0)
    rng = np.random.default_rng(0)
    theta = rng.uniform(0, 2 * np.pi, c.n_theta)
    x = rng.uniform(0, np.pi, (2, n))
    o = Observable(n)
    o.add_operator(1.0, "Z 7")
    got, _ = c.psr_full(theta, c.input_angles(x), n_sel=2, obs_list=[o])
    for j in range(2):
        want, _ = oracle_sweep(c, theta, x, [o], j=j)
        assert np.abs(got[j] - want).max() < TOL


def test_twenty_eight_qubit_split_against_whole_register():
    """28 qubits, RY + CZ ring of depth 6, transverse-field Ising terms: the split path against the tiled passes on
    the whole register (a few terms, all parameters) and the engine that sweeps the state once per gate."""
    n, depth = 28, 6
    c = create_ry_cz_ansatz(n, depth, seed=4)
    theta = np.random.default_rng(4).uniform(-np.pi, np.pi, c.n_theta)
    ia = c.input_angles(np.zeros((1, 1)))
    obs = []
    for s in ("Z 0 Z 1", "X 0", "Z 13 Z 14", "X 27", "Z 27 Z 0"):
        o = Observable(n)
        o.add_operator(1.0, s)
        obs.append(o)
    got, _ = c.psr_full(theta, ia, n_sel=1, obs_list=obs)
    with env(VQA_NO_CONE_SPLIT=1):
        c2 = create_ry_cz_ansatz(n, depth, seed=4)
        whole, _ = c2.psr_full(theta, c2.input_angles(np.zeros((1, 1))), n_sel=1, obs_list=obs)
    assert np.abs(got - whole).max() < TOL
    assert np.abs(got[0, 1:] - got[0, :1]).max() > 1e-3      # (the shifted rows do differ)
    _lib.check(_lib.lib().vqa_set_engine(4))
    try:
        c4 = create_ry_cz_ansatz(n, depth, seed=4)
        si = np.zeros(3, np.int32)
        sp = np.array([-1, 3, 150], np.int32)
        sv = np.array([0.0, 0.5 * np.pi, -0.5 * np.pi])
        ref, _ = c4.eval_batch(theta, ia, si, sp, sv, obs_list=obs)
    finally:
        _lib.lib().vqa_set_engine(0)
    assert np.abs(got[0, [0, 7, 302]] - ref).max() < TOL


def local_random_circuit(n, layers, seed, shared=False):
    """Layers of random one-qubit gates (trainable, input, constant; every kind) and random gates between ring
    neighbours: backward light cones stay a few qubits wide."""
    rng = np.random.default_rng(seed)
    c = LearningCircuit(n)
    n_par = 0
    for _ in range(layers):
        for q in range(n):
            k = int(rng.integers(8))
            ax = "XYZ"[int(rng.integers(3))]
            if k < 3:
                if shared and n_par > 2 and rng.random() < 0.2:
                    getattr(c, f"add_parametric_R{ax}_gate")(q, 0.0, share_with=int(rng.integers(n_par)),
                                                              share_with_coef=float(rng.uniform(-2, 2)))
                else:
                    getattr(c, f"add_parametric_R{ax}_gate")(q, float(rng.uniform(0, 6.28)))
                    n_par += 1
            elif k == 3:
                getattr(c, f"add_input_R{ax}_gate")(q, lambda x, q=q: x[q % len(x)] * (1 + 0.03 * q))
            elif k == 4:
                getattr(c, f"add_R{ax}_gate")(q, float(rng.uniform(-3, 3)))
            elif k == 5:
                c.add_H_gate(q)
            elif k == 6:
                getattr(c, f"add_{ax}_gate")(q)
        for q in range(int(rng.integers(2)), n, 2):
            q1 = (q + 1) % n
            r = rng.random()
            if r < 0.4:
                c.add_CZ_gate(q, q1)
            elif r < 0.6:
                c.add_CNOT_gate(q, q1)
            elif r < 0.7:
                c.add_CNOT_gate(q1, q)
            elif r < 0.8:
                c.add_CNOT_gate(q, q1)
                c.add_input_RZ_gate(q1, lambda x, q=q: x[q % len(x)] * 0.7)
                c.add_CNOT_gate(q, q1)
    return c


def local_random_observables(n, seed, n_obs=6):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_obs):
        o = Observable(n)
        for _ in range(int(rng.integers(1, 4))):
            q = int(rng.integers(n))
            s = " ".join(f"{'XYZ'[int(rng.integers(3))]} {(q + d) % n}" for d in range(int(rng.integers(1, 4))))
            o.add_operator(float(rng.uniform(-1, 1)), s)
        out.append(o)
    return out


def state_pass_counters(reset=False):
    import ctypes as C

    rw, ro = C.c_int64(0), C.c_int64(0)
    _lib.check(_lib.lib().vqa_state_pass_counters(C.byref(rw), C.byref(ro), int(reset)))
    return rw.value + ro.value


@pytest.mark.parametrize("seed", range(6))
def test_split_random_local_circuits(seed):
    """Random shallow circuits of every gate kind on 15 to 19 qubits, random local Pauli strings: sweeps, with
    every parameter driving one rotation (seeds 0-3) or some of them shared between gates (4, 5: the groups'
    circuits then run the sweep as an explicit list), and explicit lists, against the oracle on the whole
    register.  The streaming engine must not have run: the groups fit the register engine."""
    n = 15 + seed % 5
    c = local_random_circuit(n, 2 + seed % 2, 100 + seed, shared=seed >= 4)
    rng = np.random.default_rng(seed)
    theta = rng.uniform(-np.pi, np.pi, c.n_theta)
    x = rng.uniform(0, np.pi, (3, 4))
    obs = local_random_observables(n, 200 + seed)
    state_pass_counters(reset=True)
    got, gm = c.psr_full(theta, c.input_angles(x), sel=np.array([2, 0], np.int32), obs_list=obs, n_marg_bits=seed % 3)
    assert state_pass_counters() == 0
    for r, j in enumerate((2, 0)):
        want, wm = oracle_sweep(c, theta, x, obs, n_marg_bits=seed % 3, j=j)
        assert np.abs(got[r] - want).max() < TOL
        if seed % 3:
            assert np.abs(gm[r] - wm).max() < TOL
    si = np.array([1, 1, 0, 2], np.int32)
    sp = np.array([-1, 0, c.n_theta - 1, c.n_theta // 2], np.int32)
    sv = np.array([0.0, 0.3, -1.1, 0.5 * np.pi])
    want, _ = cpu_oracle.eval_batch(c, theta, c.input_angles(x), si, sp, sv, obs_list=obs_tuples(obs), n_threads=4)
    e, _ = c.eval_batch(theta, c.input_angles(x), si, sp, sv, obs_list=obs)
    assert np.abs(e - want).max() < TOL


def test_split_groups_on_the_streaming_engine():
    """21 qubits behind seven CZ rings: cones of 15 and 16 qubits, beyond the register engine -- the groups'
    reduced circuits run on the tiled passes themselves (one after the other on the caller's stream).  Against
    the whole register, and the unshifted and two shifted rows against the oracle."""
    n, depth = 21, 7
    c = create_ry_cz_ansatz(n, depth, seed=11)
    theta = np.random.default_rng(11).uniform(-np.pi, np.pi, c.n_theta)
    x = np.zeros((1, 1))
    obs = []
    for s in ("Z 0 Z 1", "X 0", "Z 10 Z 11", "X 20", "Y 5", "Z 20 Z 0"):
        o = Observable(n)
        o.add_operator(1.0, s)
        obs.append(o)
    widths = c.light_cone_groups(obs)
    assert len(widths) >= 3 and widths.min() >= 15
    got, _ = c.psr_full(theta, c.input_angles(x), n_sel=1, obs_list=obs)
    with env(VQA_NO_CONE_SPLIT=1):
        c2 = create_ry_cz_ansatz(n, depth, seed=11)
        whole, _ = c2.psr_full(theta, c2.input_angles(x), n_sel=1, obs_list=obs)
    assert np.abs(got - whole).max() < TOL
    si, sp, sv = sweep_instances(c.n_theta)
    pick = np.array([0, 2 * 80 + 1, 2 * 167 + 2])
    want, _ = cpu_oracle.eval_batch(c, theta, c.input_angles(x), si[pick], sp[pick], sv[pick], obs_list=obs_tuples(obs),
                                    n_threads=3)
    assert np.abs(got[0, pick] - want).max() < TOL
