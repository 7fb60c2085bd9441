"""GPU parity: the HIP engines, called through the C ABI, against the CPU
oracle on the same seeded inputs.  Tolerance 1e-10 absolute in fp64 (the
figure the engine is specified to)."""
import numpy as np
import pytest

from vqa_amd import LearningCircuit, Observable, _lib
from vqa_amd.qml_utils import (create_ibm_HEA_ansatz, create_jerbi_ansatz,
                               create_ry_cz_ansatz, create_ry_rz_cz_ansatz)
from oracle import cpu_oracle
from helpers import TOL, obs_tuples, random_circuit, random_observables

pytestmark = pytest.mark.gpu

ENGINES = {"lds": 1, "reg": 2, "hbm": 3, "pergate": 4}


@pytest.fixture(autouse=True)
def _reset_engine():
    yield
    _lib.lib().vqa_set_engine(0)


def set_engine(name):
    _lib.check(_lib.lib().vqa_set_engine(ENGINES[name] if name else 0))


def compare(c, theta, x, obs_list, marg, si=None, sp=None, sv=None, device_path=False):
    ia = c.input_angles(x)
    want_e, want_m = cpu_oracle.eval_batch(c, theta, ia, si, sp, sv, obs_list=obs_tuples(obs_list),
                                           n_marg_bits=marg)
    inputs = c.upload_inputs(x) if device_path else ia
    got_e, got_m = c.eval_batch(theta, inputs, si, sp, sv, obs_list=obs_list, n_marg_bits=marg)
    if device_path:
        got_e = got_e.cpu().numpy()
        got_m = None if got_m is None else got_m.cpu().numpy()
    assert got_e.shape == want_e.shape
    if want_e.size:
        assert np.abs(got_e - want_e).max() < TOL
    if marg:
        assert np.abs(got_m - want_m).max() < TOL
        assert np.abs(got_m.sum(axis=1) - 1).max() < TOL


@pytest.mark.parametrize("n", [4, 5, 6, 7, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("xy", [False, True])
def test_random_circuits_register_engine(n, xy):
    """Engine 2 (register-blocked); with X/Y Pauli terms the call is served by
    the gate-by-gate engine behind the same handle."""
    set_engine("reg")
    c = random_circuit(LearningCircuit, n, 70, seed=300 + n)
    rng = np.random.default_rng(50 + n)
    theta = rng.uniform(-3, 3, c.n_theta)
    x = rng.uniform(-1, 1, (9, 3))
    E = 37
    si = rng.integers(0, 9, E).astype(np.int32)
    sp = rng.integers(-1, c.n_theta, E).astype(np.int32)
    sv = rng.choice([0.5 * np.pi, -0.5 * np.pi, 0.3], E)
    obs = random_observables(n, seed=n, xy=xy)
    compare(c, theta, x, obs, 2, si, sp, sv)
    compare(c, theta, x, obs, 2, si, sp, sv, device_path=True)


@pytest.mark.parametrize("engine", ["lds", "hbm"])
@pytest.mark.parametrize("n", [1, 2, 3, 5, 6, 8, 9, 10, 12])
def test_random_circuits_all_gate_kinds(engine, n):
    set_engine(engine)
    c = random_circuit(LearningCircuit, n, 60, seed=100 + n)
    rng = np.random.default_rng(n)
    theta = rng.uniform(-3, 3, c.n_theta)
    x = rng.uniform(-1, 1, (7, 3))
    E = 19
    si = rng.integers(0, 7, E).astype(np.int32)
    sp = rng.integers(-1, c.n_theta, E).astype(np.int32)
    sv = rng.choice([0.5 * np.pi, -0.5 * np.pi, 0.3], E)
    obs = random_observables(n, seed=n, xy=True)
    compare(c, theta, x, obs, min(n, 2), si, sp, sv)
    compare(c, theta, x, obs, min(n, 2), si, sp, sv, device_path=True)


@pytest.mark.parametrize("engine,n,d", [("reg", 4, 2), ("reg", 6, 5), ("reg", 9, 2), ("reg", 11, 2), ("reg", 12, 2),
                                        ("reg", 12, 10), ("reg", 13, 1), ("reg", 14, 1),
                                        ("lds", 4, 2), ("lds", 6, 5), ("lds", 12, 2), ("lds", 13, 1),
                                        ("hbm", 6, 2), ("hbm", 14, 1), ("hbm", 16, 1), ("hbm", 17, 2), ("pergate", 14, 1)])
def test_ibm_hea_full_parameter_shift(engine, n, d):
    set_engine(engine)
    c = create_ibm_HEA_ansatz(n, d, seed=0)
    rng = np.random.default_rng(n + d)
    theta = rng.uniform(0, 2 * np.pi, c.n_theta)
    B = 3 if n >= 12 else 5
    x = rng.uniform(0, np.pi, (B, n))
    ia = c.input_angles(x)
    z0 = Observable(n); z0.add_operator(1.0, "Z 0")
    sel = np.array([B - 1, 0], np.int32)
    si, sp, sv = cpu_oracle.psr_lists(2, c.n_theta, sel=sel)
    want_e, want_m = cpu_oracle.eval_batch(c, theta, ia, si, sp, sv, obs_list=obs_tuples([z0]), n_marg_bits=2)
    K = 2 * c.n_theta + 1
    # host-pointer path
    got_e, got_m = c.psr_full(theta, ia, sel=sel, obs_list=[z0], n_marg_bits=2)
    assert got_e.shape == (2, K, 1)
    assert np.abs(got_e.reshape(-1, 1) - want_e).max() < TOL
    assert np.abs(got_m.reshape(-1, 4) - want_m).max() < TOL
    # device-pointer path, inputs resident
    d_in = c.upload_inputs(x)
    got_e2, got_m2 = c.psr_full(theta, d_in, sel=sel, obs_list=[z0], n_marg_bits=2)
    assert np.abs(got_e2.cpu().numpy().reshape(-1, 1) - want_e).max() < TOL
    assert np.abs(got_m2.cpu().numpy().reshape(-1, 4) - want_m).max() < TOL


@pytest.mark.parametrize("engine", ["reg", "lds", "hbm"])
@pytest.mark.parametrize("builder,n,d", [(create_jerbi_ansatz, 5, 2), (create_jerbi_ansatz, 9, 1), (create_jerbi_ansatz, 12, 1),
                                         (create_ry_rz_cz_ansatz, 4, 3), (create_ry_cz_ansatz, 10, 3)])
def test_other_ansaetze(engine, builder, n, d):
    set_engine(engine)
    c = builder(n, d, seed=2)
    rng = np.random.default_rng(7)
    theta = rng.uniform(0, 2 * np.pi, c.n_theta)
    x = rng.uniform(0, np.pi, (6, n))
    obs = random_observables(n, seed=3, xy=(builder is create_ry_cz_ansatz and engine != "reg"))
    compare(c, theta, x, obs, 2)


def test_classifier_observables_and_marginal():
    c = create_ry_rz_cz_ansatz(4, 3, seed=0)
    rng = np.random.default_rng(0)
    theta = rng.uniform(0, 2 * np.pi, c.n_theta)
    x = rng.uniform(0, np.pi, (150, 4))
    obs = []
    for terms in ([(1 / 3, ""), (1 / 3, "Z 0 Z 1"), (1 / 6, "Z 0"), (1 / 6, "Z 1")],
                  [(1 / 3, ""), (-1 / 3, "Z 0"), (1 / 6, "Z 1"), (-1 / 6, "Z 0 Z 1")],
                  [(1 / 3, ""), (-1 / 3, "Z 1"), (1 / 6, "Z 0"), (-1 / 6, "Z 0 Z 1")]):
        o = Observable(4)
        for coef, s in terms:
            o.add_operator(coef, s)
        obs.append(o)
    e, m = c.eval_batch(theta, c.input_angles(x), obs_list=obs, n_marg_bits=2)
    assert np.abs(e - (m[:, :3] + m[:, 3:4] / 3)).max() < TOL
    compare(c, theta, x, obs, 2)


def test_statevector_and_run_shim():
    c = create_ibm_HEA_ansatz(5, 1, seed=4)
    x = np.array([0.3, 1.2, 2.0, 0.7, 2.9])
    theta = np.array(c.get_parameters())
    want = cpu_oracle.statevector(c, theta, c.input_row(x))
    state = c.run(x)
    assert np.abs(state.get_vector() - want).max() < TOL
    z0 = Observable(5); z0.add_operator(1.0, "Z 0")
    ev = z0.get_expectation_value(state)
    assert abs(ev.real - (np.abs(want[0::2]) ** 2 - np.abs(want[1::2]) ** 2).sum()) < TOL
    samples = np.array(state.sampling(2000, seed=1))
    assert samples.min() >= 0 and samples.max() < 32
    assert abs(((-1.0) ** (samples % 2)).mean() - ev.real) < 0.1


def test_empty_batch_and_errors():
    c = create_ibm_HEA_ansatz(3, 1)
    theta = np.array(c.get_parameters())
    ia = c.input_angles(np.zeros((2, 3)))
    z0 = Observable(3); z0.add_operator(1.0, "Z 0")
    e, _ = c.eval_batch(theta, ia, sample_idx=np.zeros(0, np.int32), obs_list=[z0])
    assert e.shape == (0, 1)
    with pytest.raises(_lib.VqaError):
        c.eval_batch(theta, ia, sample_idx=np.array([5], np.int32), obs_list=[z0])
    with pytest.raises(_lib.VqaError):
        c.eval_batch(theta, ia, sample_idx=np.array([0], np.int32), shift_param=np.array([99], np.int32),
                     shift_val=np.array([1.0]), obs_list=[z0])
    bad = Observable(4); bad.add_operator(1.0, "Z 3")
    with pytest.raises(_lib.VqaError):
        c.eval_batch(theta, ia, obs_list=[bad])


def test_large_batch_linearity_property():
    """Size-independent check at a batch the oracle would not finish quickly:
    sum_c p_c = 1 and <Z0> = m0 - m1 + m2 - m3 for every instance."""
    set_engine("lds")
    c = create_ibm_HEA_ansatz(6, 5, seed=0)
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (4096, 1))
    d_in = c.upload_inputs(x)
    z0 = Observable(6); z0.add_operator(1.0, "Z 0")
    theta = np.array(c.get_parameters())
    e, m = c.eval_batch(theta, d_in, obs_list=[z0], n_marg_bits=2)
    e, m = e.cpu().numpy(), m.cpu().numpy()
    assert np.abs(m.sum(axis=1) - 1).max() < TOL
    assert np.abs(e[:, 0] - (m[:, 0] - m[:, 1] + m[:, 2] - m[:, 3])).max() < TOL
    sub = rng.choice(4096, 64, replace=False)
    want, _ = cpu_oracle.eval_batch(c, theta, c.input_angles(x[sub]), obs_list=obs_tuples([z0]))
    assert np.abs(e[sub] - want).max() < TOL


def test_headline_sweep_against_explicit_list_and_oracle():
    """12-qubit depth-10 IBM-HEA (the benchmark circuit): the sweep kernel (checkpoints,
    pair evaluation) against the explicit-list kernel on every instance of 48 samples,
    and against the oracle on a subset."""
    c = create_ibm_HEA_ansatz(12, 10, seed=0)
    rng = np.random.default_rng(5)
    theta = rng.uniform(0, 2 * np.pi, c.n_theta)
    B = 48
    x = rng.uniform(0, np.pi, (B, 12))
    d_in = c.upload_inputs(x)
    z0 = Observable(12); z0.add_operator(1.0, "Z 0")
    zz = Observable(12); zz.add_operator(0.5, "Z 1 Z 7"); zz.add_operator(-0.25, "")
    K = 2 * c.n_theta + 1
    e, m = c.psr_full(theta, d_in, n_sel=B, obs_list=[z0, zz], n_marg_bits=2)
    e, m = e.cpu().numpy(), m.cpu().numpy()
    assert np.isfinite(e).all()
    assert np.abs(m.sum(axis=2) - 1).max() < TOL
    assert np.abs(e[:, :, 0] - (m[:, :, 0] - m[:, :, 1] + m[:, :, 2] - m[:, :, 3])).max() < TOL
    si, sp, sv = cpu_oracle.psr_lists(B, c.n_theta)
    e2, m2 = c.eval_batch(theta, d_in, si, sp, sv, obs_list=[z0, zz], n_marg_bits=2)
    assert np.abs(e2.cpu().numpy().reshape(B, K, 2) - e).max() < TOL
    assert np.abs(m2.cpu().numpy().reshape(B, K, 4) - m).max() < TOL
    pick = rng.choice(B * K, 300, replace=False)
    want_e, want_m = cpu_oracle.eval_batch(c, theta, c.input_angles(x), si[pick], sp[pick], sv[pick],
                                           obs_list=obs_tuples([z0, zz]), n_marg_bits=2)
    assert np.abs(e.reshape(B * K, 2)[pick] - want_e).max() < TOL
    assert np.abs(m.reshape(B * K, 4)[pick] - want_m).max() < TOL


def test_headline_full_batch_properties():
    """The benchmark's full size (65536 samples x 793 instances): size-independent properties of
    every instance -- probabilities sum to one, <Z0> equals the signed sum of the marginal, the
    unshifted column does not depend on the batch around it -- and a seeded subset against the oracle."""
    import torch

    c = create_ibm_HEA_ansatz(12, 10, seed=0)
    rng = np.random.default_rng(2)
    B = 65536
    x = rng.uniform(0, np.pi, (B, 12))
    theta = np.random.default_rng(0).uniform(0, 2 * np.pi, c.n_theta)
    d_in = c.upload_inputs(x)
    z0 = Observable(12); z0.add_operator(1.0, "Z 0")
    e, m = c.psr_full(theta, d_in, n_sel=B, obs_list=[z0], n_marg_bits=2)
    assert bool(torch.isfinite(e).all()) and bool(torch.isfinite(m).all())
    assert float((m.sum(dim=2) - 1).abs().max()) < TOL
    assert float((e[:, :, 0] - (m[:, :, 0] - m[:, :, 1] + m[:, :, 2] - m[:, :, 3])).abs().max()) < TOL
    assert float(m.min()) > -TOL
    sub = np.sort(rng.choice(B, 40, replace=False)).astype(np.int32)
    e2, _ = c.psr_full(theta, d_in, sel=sub, obs_list=[z0])
    assert float((e2 - e[torch.from_numpy(sub.astype(np.int64)).cuda()]).abs().max()) < TOL
    K = 2 * c.n_theta + 1
    si, sp, sv = cpu_oracle.psr_lists(len(sub), c.n_theta, sel=sub)
    pick = rng.choice(len(si), 200, replace=False)
    want, _ = cpu_oracle.eval_batch(c, theta, c.input_angles(x), si[pick], sp[pick], sv[pick], obs_list=obs_tuples([z0]))
    assert np.abs(e2.cpu().numpy().reshape(len(sub) * K, 1)[pick] - want).max() < TOL


def test_regressor_config_full_batch_properties():
    """The sin(x) regressor's configuration at full size (6 qubits, depth 5, 4096 samples, one feature):
    the grouped sweep of the whole mini-batch is normalised, consistent between expectation and
    marginal, equal to the sweep of a subset, to the same instances as an explicit list, to the
    oracle on a seeded sample of instances, and its shifted rows obey the parameter-shift identity
    against a central difference."""
    import torch
    c = create_ibm_HEA_ansatz(6, 5, seed=0)
    rng = np.random.default_rng(1)
    B = 4096
    theta = rng.uniform(0, 2 * np.pi, c.n_theta)
    x = rng.uniform(-1, 1, (B, 1))
    d_in = c.upload_inputs(x)
    z0 = Observable(6); z0.add_operator(1.0, "Z 0")
    e, m = c.psr_full(theta, d_in, n_sel=B, obs_list=[z0], n_marg_bits=2)
    K = 2 * c.n_theta + 1
    assert e.shape == (B, K, 1)
    assert bool(torch.isfinite(e).all()) and bool(torch.isfinite(m).all())
    assert float((m.sum(dim=2) - 1).abs().max()) < TOL
    assert float((e[:, :, 0] - (m[:, :, 0] - m[:, :, 1] + m[:, :, 2] - m[:, :, 3])).abs().max()) < TOL
    sub = np.sort(rng.choice(B, 64, replace=False)).astype(np.int32)
    e2, _ = c.psr_full(theta, d_in, sel=sub, obs_list=[z0])
    assert float((e2 - e[torch.from_numpy(sub.astype(np.int64)).cuda()]).abs().max()) < TOL
    si, sp, sv = cpu_oracle.psr_lists(len(sub), c.n_theta, sel=sub)
    e3, _ = c.eval_batch(theta, d_in, si, sp, sv, obs_list=[z0])
    assert float((e3.reshape(len(sub), K, 1) - e2).abs().max()) < TOL
    pick = rng.choice(len(si), 400, replace=False)
    want, _ = cpu_oracle.eval_batch(c, theta, c.input_angles(x), si[pick], sp[pick], sv[pick], obs_list=obs_tuples([z0]))
    assert np.abs(e2.cpu().numpy().reshape(len(sub) * K, 1)[pick] - want).max() < TOL
    e_host = e[:8].cpu().numpy()[:, :, 0]
    for i in (0, 17, c.n_theta - 1):
        h = 1e-5
        fd, _ = c.eval_batch(theta, d_in, np.repeat(np.arange(8, dtype=np.int32), 2), np.full(16, i, np.int32),
                             np.tile([h, -h], 8), obs_list=[z0])
        fd = fd.cpu().numpy().reshape(8, 2)
        assert np.abs(0.5 * (e_host[:, 2 * i + 1] - e_host[:, 2 * i + 2]) - (fd[:, 0] - fd[:, 1]) / (2 * h)).max() < 1e-8


def test_twenty_qubit_sweep_against_explicit_list():
    """The 20-qubit configuration's circuit (depth 8, 900 gates, 540 parameters): the streaming
    engine's full sweep of one sample against its explicit-list form on a seeded subset of the 1081
    instances, the parameter-shift identity against a central difference, and normalisation."""
    c = create_ibm_HEA_ansatz(20, 8, seed=0)
    rng = np.random.default_rng(3)
    theta = rng.uniform(0, 2 * np.pi, c.n_theta)
    x = rng.uniform(0, np.pi, (1, 20))
    d_in = c.upload_inputs(x)
    z0 = Observable(20); z0.add_operator(1.0, "Z 0")
    K = 2 * c.n_theta + 1
    e, m = c.psr_full(theta, d_in, n_sel=1, obs_list=[z0], n_marg_bits=2)
    e, m = e.cpu().numpy().reshape(K), m.cpu().numpy().reshape(K, 4)
    assert np.abs(m.sum(axis=1) - 1).max() < TOL
    assert np.abs(e - (m[:, 0] - m[:, 1] + m[:, 2] - m[:, 3])).max() < TOL
    si, sp, sv = cpu_oracle.psr_lists(1, c.n_theta)
    pick = np.sort(rng.choice(K, 24, replace=False))
    e2, m2 = c.eval_batch(theta, d_in, si[pick], sp[pick], sv[pick], obs_list=[z0], n_marg_bits=2)
    assert np.abs(e2.cpu().numpy()[:, 0] - e[pick]).max() < TOL
    assert np.abs(m2.cpu().numpy() - m[pick]).max() < TOL
    for i in (5, 300, c.n_theta - 1):
        h = 1e-5
        fd, _ = c.eval_batch(theta, d_in, np.zeros(2, np.int32), np.array([i, i], np.int32), np.array([h, -h]), obs_list=[z0])
        fd = fd.cpu().numpy()[:, 0]
        assert abs(0.5 * (e[2 * i + 1] - e[2 * i + 2]) - (fd[0] - fd[1]) / (2 * h)) < 1e-8


def test_sweep_with_shared_and_scaled_parameters():
    """Parameters that enter several gates, or with a coefficient: the sweep falls back
    to explicit shifts or uses the scaled half-angle, and still matches the oracle."""
    set_engine("reg")
    c = LearningCircuit(10)
    for q in range(10):
        c.add_H_gate(q)
    p0 = c.add_parametric_RY_gate(0, 0.3)
    c.add_parametric_RX_gate(1, 0.0, share_with=p0, share_with_coef=-1.5)   # same parameter, other entry
    for q in range(2, 10):
        c.add_parametric_RZ_gate(q, 0.1 * q)
        c.add_parametric_RX_gate(q, 0.2 * q)
    for q in range(10):
        c.add_CNOT_gate(q, (q + 1) % 10)
    for q in range(10):
        c.add_input_RY_gate(q, lambda x, q=q: x[q % len(x)])
        c.add_parametric_RY_gate(q, 0.05 * q)
    for q in range(0, 10, 2):
        c.add_CZ_gate(q, q + 1)
    for q in range(10):
        c.add_parametric_RX_gate(q, 0.3 + 0.01 * q)
    rng = np.random.default_rng(3)
    theta = rng.uniform(0, 2 * np.pi, c.n_theta)
    x = rng.uniform(0, np.pi, (4, 3))
    ia = c.input_angles(x)
    obs = random_observables(10, seed=9)
    si, sp, sv = cpu_oracle.psr_lists(4, c.n_theta)
    want_e, want_m = cpu_oracle.eval_batch(c, theta, ia, si, sp, sv, obs_list=obs_tuples(obs), n_marg_bits=3)
    got_e, got_m = c.psr_full(theta, ia, obs_list=obs, n_marg_bits=3)
    assert np.abs(got_e.reshape(want_e.shape) - want_e).max() < TOL
    assert np.abs(got_m.reshape(want_m.shape) - want_m).max() < TOL


@pytest.mark.parametrize("n", [13, 14])
def test_thirteen_fourteen_qubits_automatic_engine(n):
    """13 and 14 qubits, automatic engine choice: the register engine (512 / 1024 threads per state)
    for Z-basis calls, the streaming engine's tiles (two / four per state) when the call has X/Y terms."""
    c = create_ibm_HEA_ansatz(n, 2, seed=3)
    rng = np.random.default_rng(13)
    theta = rng.uniform(0, 2 * np.pi, c.n_theta)
    x = rng.uniform(0, np.pi, (2, n))
    ia = c.input_angles(x)
    z0 = Observable(n); z0.add_operator(1.0, "Z 0"); z0.add_operator(0.5, "Z 5 Z 12")
    si, sp, sv = cpu_oracle.psr_lists(2, c.n_theta)
    want_e, want_m = cpu_oracle.eval_batch(c, theta, ia, si, sp, sv, obs_list=obs_tuples([z0]), n_marg_bits=2)
    got_e, got_m = c.psr_full(theta, ia, obs_list=[z0], n_marg_bits=2)
    assert np.abs(got_e.reshape(want_e.shape) - want_e).max() < TOL
    assert np.abs(got_m.reshape(want_m.shape) - want_m).max() < TOL
    compare(c, theta, x, [z0], 2, si[:9], sp[:9], sv[:9])
    xy = Observable(n); xy.add_operator(0.4, "X 2 Y 9"); xy.add_operator(0.6, "Z 1"); xy.add_operator(-0.3, "X 12")
    compare(c, theta, x, [xy, z0], 2, si[:9], sp[:9], sv[:9])
    got_x, _ = c.psr_full(theta, ia, obs_list=[xy])
    want_x, _ = cpu_oracle.eval_batch(c, theta, ia, si, sp, sv, obs_list=obs_tuples([xy]))
    assert np.abs(got_x.reshape(want_x.shape) - want_x).max() < TOL


def test_twenty_qubit_streaming_engine():
    """The 20-qubit configuration's shape on the HBM engine (automatic choice)."""
    c = create_ibm_HEA_ansatz(20, 1, seed=1)
    rng = np.random.default_rng(11)
    theta = rng.uniform(0, 2 * np.pi, c.n_theta)
    x = rng.uniform(0, np.pi, (2, 20))
    z0 = Observable(20); z0.add_operator(1.0, "Z 0")
    zs = Observable(20); zs.add_operator(0.7, "Z 3 Z 19"); zs.add_operator(0.2, "X 5")
    si = np.array([0, 1, 1], np.int32)
    sp = np.array([-1, 7, c.n_theta - 1], np.int32)
    sv = np.array([0.0, 0.5 * np.pi, -0.5 * np.pi])
    compare(c, theta, x, [z0, zs], 2, si, sp, sv)


def test_pauli_sum_ring_hamiltonian():
    """VQE-shaped case: RY + CZ ring ansatz, transverse-field Ising Pauli sum with X terms."""
    n = 16
    c = create_ry_cz_ansatz(n, 6, seed=4)
    rng = np.random.default_rng(2)
    theta = rng.uniform(-np.pi, np.pi, c.n_theta)
    h = Observable(n)
    for i in range(n):
        h.add_operator(1.0, f"Z {i} Z {(i + 1) % n}")
        h.add_operator(1.0, f"X {i}")
    si, sp, sv = cpu_oracle.psr_lists(1, c.n_theta)
    keep = rng.choice(len(si), 12, replace=False)
    keep[0] = 0
    ia = c.input_angles(np.zeros((1, 1)))
    want, _ = cpu_oracle.eval_batch(c, theta, ia, si[keep], sp[keep], sv[keep], obs_list=obs_tuples([h]))
    got, _ = c.eval_batch(theta, ia, si[keep], sp[keep], sv[keep], obs_list=[h])
    assert np.abs(got - want).max() < 1e-9   # 32 terms of order one: absolute 1e-10 per term


@pytest.mark.parametrize("tile12", [False, True])
@pytest.mark.parametrize("n,d", [(14, 2), (15, 1)])
def test_streaming_sweep_with_pauli_sums(n, d, tile12, monkeypatch):
    """Full sweeps on the HBM engine with several observables at once: Z-strings of different
    observables batched together, pure-X terms measured behind a Hadamard layer, mixed X/Y/Z
    strings one by one; every instance against the oracle.  Both builds of the tiled passes
    (11- and 12-qubit tiles)."""
    if tile12:
        monkeypatch.setenv("VQA_TILE12", "1")
    c = create_ry_cz_ansatz(n, d, seed=4)
    rng = np.random.default_rng(n)
    theta = rng.uniform(-np.pi, np.pi, c.n_theta)
    tfim = Observable(n)
    for i in range(n):
        tfim.add_operator(1.0, f"Z {i} Z {(i + 1) % n}")
        tfim.add_operator(0.7, f"X {i}")
    mixed = Observable(n)
    mixed.add_operator(0.5, "X 0 X 5")
    if n == 15:
        # a pure-X string across two passes of the Hadamard layer: the layer's passes then run one after
        # the other (only the last one measures) instead of side by side
        mixed.add_operator(0.25, f"X 0 X {n - 1}")
    mixed.add_operator(-0.3, f"Y 2 Z {n - 1}")
    mixed.add_operator(0.2, f"X 1 Z 3 Y {n - 2}")
    mixed.add_operator(0.1, "")
    z0 = Observable(n); z0.add_operator(1.0, "Z 0")
    obs = [tfim, mixed, z0]
    ia = c.input_angles(np.zeros((1, 1)))
    si, sp, sv = cpu_oracle.psr_lists(1, c.n_theta)
    want_e, want_m = cpu_oracle.eval_batch(c, theta, ia, si, sp, sv, obs_list=obs_tuples(obs), n_marg_bits=2)
    got_e, got_m = c.psr_full(theta, ia, obs_list=obs, n_marg_bits=2)
    assert np.abs(got_e.reshape(want_e.shape) - want_e).max() < 1e-9   # sums of up to 2n terms of order one
    assert np.abs(got_m.reshape(want_m.shape) - want_m).max() < TOL
    # the explicit instance list takes the same measurement path without the pairs
    got_e2, _ = c.eval_batch(theta, ia, si[:7], sp[:7], sv[:7], obs_list=obs)
    assert np.abs(got_e2 - want_e[:7]).max() < 1e-9


@pytest.mark.parametrize("n,d,builder", [(8, 2, create_ibm_HEA_ansatz), (12, 2, create_ibm_HEA_ansatz), (14, 2, create_ry_cz_ansatz), (15, 2, create_ry_cz_ansatz)])
def test_parameter_subset_sweep(n, d, builder):
    """vqa_psr_params_dev: the rows of the requested parameters (dead ones included) and the
    unshifted row equal the full sweep's; register engine and streaming engine."""
    c = builder(n, d, seed=1)
    rng = np.random.default_rng(n)
    theta = rng.uniform(-3, 3, c.n_theta)
    x = rng.uniform(0, np.pi, (2, n))
    d_in = c.upload_inputs(x)
    z0 = Observable(n); z0.add_operator(1.0, "Z 0"); z0.add_operator(0.3, "Z 1 Z 2")
    full_e, full_m = c.psr_full(theta, d_in, obs_list=[z0], n_marg_bits=2)
    mine = np.arange(1, c.n_theta, 3)
    e, m = c.psr_full(theta, d_in, obs_list=[z0], n_marg_bits=2, params=mine)
    rows = np.concatenate([[0], 2 * mine + 1, 2 * mine + 2])
    assert float((e[:, rows] - full_e[:, rows]).abs().max()) < TOL
    assert float((m[:, rows] - full_m[:, rows]).abs().max()) < TOL
    assert c.dead_parameters([z0], 2)[mine].any()          # the subset holds parameters outside the light cone too
    # host arrays in, host arrays out
    eh, _ = c.psr_full(theta, c.input_angles(x), obs_list=[z0], params=mine[:2])
    assert isinstance(eh, np.ndarray)
    assert np.abs(eh[:, [0, 2 * mine[0] + 1, 2 * mine[1] + 2]] - full_e.cpu().numpy()[:, [0, 2 * mine[0] + 1, 2 * mine[1] + 2]]).max() < TOL


@pytest.mark.parametrize("tile12", [False, True])
@pytest.mark.parametrize("cap", [None, 44, 30, 12])
def test_streaming_sweep_in_chunks_with_late_branching(cap, tile12, monkeypatch):
    """IBM-HEA on 16 qubits, depth 3, on the HBM engine: the parameters' states branch off the running
    state at the first sweep of their pass they can change.  With the number of states in HBM capped
    the sweep runs in several chunks of parameters: with the branching points of all passes kept (left by
    the sample's unshifted run, cap 44) or those of one pass at a time (left by a second unshifted state
    that runs beside the chunks, which share them when they branch in the same pass, cap 30); a cap that
    leaves no room for branching points runs the earlier form (cap 12).  Every row against the oracle, and one sample's rows again from a sweep of two samples."""
    n, d = 16, 3
    if cap is not None:
        monkeypatch.setenv("VQA_HBM_MAX_STATES", str(cap))
    if tile12:
        monkeypatch.setenv("VQA_TILE12", "1")   # the build with 12-qubit tiles (the library's choice from 26 qubits on)
    c = create_ibm_HEA_ansatz(n, d, seed=2)
    rng = np.random.default_rng(16)
    theta = rng.uniform(0, 2 * np.pi, c.n_theta)
    x = rng.uniform(0, np.pi, (2, n))
    ia = c.input_angles(x)
    z = Observable(n); z.add_operator(1.0, "Z 0"); z.add_operator(-0.4, "Z 3 Z 9"); z.add_operator(0.2, f"Z {n - 1}")
    si, sp, sv = cpu_oracle.psr_lists(1, c.n_theta)
    want_e, want_m = cpu_oracle.eval_batch(c, theta, ia[1:], si, sp, sv, obs_list=obs_tuples([z]), n_marg_bits=2)
    got_e, got_m = c.psr_full(theta, ia, obs_list=[z], n_marg_bits=2)
    assert np.abs(got_e[1].reshape(want_e.shape) - want_e).max() < TOL
    assert np.abs(got_m[1].reshape(want_m.shape) - want_m).max() < TOL
    one_e, one_m = c.psr_full(theta, ia[:1], obs_list=[z], n_marg_bits=2)
    assert np.abs(one_e[0] - got_e[0]).max() < TOL and np.abs(one_m[0] - got_m[0]).max() < TOL
    # a subset of the parameters (other chunks, other branching points in use)
    mine = np.arange(2, c.n_theta, 5)
    sub_e, sub_m = c.psr_full(theta, ia, obs_list=[z], n_marg_bits=2, params=mine)
    rows = np.concatenate([[0], 2 * mine + 1, 2 * mine + 2])
    assert np.abs(sub_e[:, rows] - got_e[:, rows]).max() < TOL and np.abs(sub_m[:, rows] - got_m[:, rows]).max() < TOL


@pytest.mark.parametrize("tile12", [False, True])
@pytest.mark.parametrize("n,seed", [(13, 1), (14, 2), (15, 3), (15, 4), (16, 5)])
def test_streaming_sweep_random_circuits(n, seed, tile12, monkeypatch):
    """Full sweeps of random circuits on the HBM engine (tiled passes with lone CNOTs, constant and input
    gates of every kind between the parameters' rotations, which makes the parameters branch at many
    different sweeps), both tile shapes; every row of two samples against the oracle."""
    from helpers import random_sweep_circuit
    if tile12:
        monkeypatch.setenv("VQA_TILE12", "1")
    set_engine("hbm")
    c = random_sweep_circuit(LearningCircuit, n, 90, seed=seed)
    rng = np.random.default_rng(seed)
    theta = rng.uniform(-3, 3, c.n_theta)
    x = rng.uniform(-1, 1, (2, 3))
    ia = c.input_angles(x)
    obs = random_observables(n, seed=seed, n_obs=2, max_terms=3)
    si, sp, sv = cpu_oracle.psr_lists(2, c.n_theta)
    want_e, want_m = cpu_oracle.eval_batch(c, theta, ia, si, sp, sv, obs_list=obs_tuples(obs), n_marg_bits=2)
    got_e, got_m = c.psr_full(theta, ia, obs_list=obs, n_marg_bits=2)
    assert np.abs(np.asarray(got_e).reshape(want_e.shape) - want_e).max() < TOL
    assert np.abs(np.asarray(got_m).reshape(want_m.shape) - want_m).max() < TOL


@pytest.mark.parametrize("lanes,streams", [(None, None), (2, 2), (3, 1), (1, 4)])
def test_streaming_sweep_groups_of_samples(lanes, streams, monkeypatch):
    """Several samples per launch on the HBM engine (a lane of the grid per sample) and groups of samples on
    several streams: seven samples in the library's own grouping, in groups of two on two streams (the last
    group holds one), in groups of three on one stream and one by one on four streams; selected sample
    indices out of order, X terms behind the Hadamard layer; every row against the oracle."""
    from helpers import random_sweep_circuit
    n = 14
    if lanes is not None:
        monkeypatch.setenv("VQA_HBM_GRID_LANES", str(lanes))
        monkeypatch.setenv("VQA_HBM_STREAMS", str(streams))
    set_engine("hbm")
    c = random_sweep_circuit(LearningCircuit, n, 70, seed=21, n_theta=9)
    rng = np.random.default_rng(21)
    theta = rng.uniform(-3, 3, c.n_theta)
    x = rng.uniform(-1, 1, (9, 3))
    ia = c.input_angles(x)
    obs = random_observables(n, seed=5, n_obs=2, max_terms=3)
    tf = Observable(n)
    for i in range(4):
        tf.add_operator(0.5 + 0.1 * i, f"X {i}")
    obs = obs + [tf]
    sel = np.array([8, 0, 3, 3, 7, 1, 5], np.int32)
    si, sp, sv = cpu_oracle.psr_lists(len(sel), c.n_theta, sel=sel)
    want_e, want_m = cpu_oracle.eval_batch(c, theta, ia, si, sp, sv, obs_list=obs_tuples(obs), n_marg_bits=2)
    got_e, got_m = c.psr_full(theta, ia, sel=sel, obs_list=obs, n_marg_bits=2)
    assert np.abs(np.asarray(got_e).reshape(want_e.shape) - want_e).max() < TOL
    assert np.abs(np.asarray(got_m).reshape(want_m.shape) - want_m).max() < TOL


def _edge_circuit(n, shared):
    """Parameters with a coefficient, a parameter that drives no gate, (optionally) one shared by two
    gates, CNOT sandwiches, input gates, rotations of every axis on low, middle and top qubits."""
    from vqa_amd.circuit import SRC_THETA
    c = LearningCircuit(n)
    for q in range(n):
        c.add_H_gate(q)
        c.add_input_RZ_gate(q, lambda x, q=q: x[q % len(x)])
    for q in range(n - 1):
        c.add_CNOT_gate(q, q + 1)
        c.add_input_RZ_gate(q + 1, lambda x, q=q: x[q % len(x)] * x[(q + 1) % len(x)])
        c.add_CNOT_gate(q, q + 1)
    for q in range(n):
        c.add_parametric_RY_gate(q, 0.1 * q)
        c.add_parametric_RZ_gate(q, 0.2)
    c._theta.append(0.7)                                   # parameter 2n: drives nothing
    for q in range(0, n, 2):
        c.add_CZ_gate(q, (q + 1) % n)
    first = None
    for q in range(n):
        pid = c.add_parametric_RX_gate(q, 0.3)
        first = pid if first is None else first
    if shared:
        c.add_parametric_RY_gate(n - 1, 0.0, share_with=first, share_with_coef=-0.5)
    # coefficients other than one on single-gate parameters
    ops = c._ops
    for k, op in enumerate(ops):
        if op[3] == SRC_THETA and op[4] in (1, 2 * n + 2):
            ops[k] = op[:6] + (2.0 if op[4] == 1 else -1.5, op[7])
    c.add_CNOT_gate(n - 1, 0)
    c.add_parametric_RY_gate(0, 0.4)
    return c


@pytest.mark.parametrize("tile12", [False, True])
@pytest.mark.parametrize("shared", [False, True])
def test_streaming_sweep_edge_cases(shared, tile12, monkeypatch):
    """The streaming engine's sweep (one moved state per parameter) on a circuit with scaled
    parameters and one that drives no gate; with a parameter shared by two gates the call is
    served by the explicit list.  A sample index beyond the batch gives NaN rows; an observable
    without terms gives zeros."""
    n = 14
    if tile12:
        monkeypatch.setenv("VQA_TILE12", "1")
    c = _edge_circuit(n, shared)
    rng = np.random.default_rng(17)
    theta = rng.uniform(-3, 3, c.n_theta)
    x = rng.uniform(0, np.pi, (3, 4))
    ia = c.input_angles(x)
    o1 = Observable(n); o1.add_operator(0.6, "Z 0 Z 13"); o1.add_operator(-0.2, "X 5"); o1.add_operator(0.3, "")
    empty = Observable(n)
    o2 = Observable(n); o2.add_operator(1.0, "Z 7")
    obs = [o1, empty, o2]
    sel = np.array([2, 0], np.int32)
    si, sp, sv = cpu_oracle.psr_lists(2, c.n_theta, sel=sel)
    want_e, want_m = cpu_oracle.eval_batch(c, theta, ia, si, sp, sv, obs_list=obs_tuples(obs), n_marg_bits=3)
    got_e, got_m = c.psr_full(theta, ia, sel=sel, obs_list=obs, n_marg_bits=3)
    assert np.abs(got_e.reshape(want_e.shape) - want_e).max() < TOL
    assert np.abs(got_m.reshape(want_m.shape) - want_m).max() < TOL
    assert np.all(got_e[:, :, 1] == 0.0)
    K = 2 * c.n_theta + 1
    unused = 2 * n
    assert np.array_equal(got_e[:, 2 * unused + 1], got_e[:, 0]) and np.array_equal(got_e[:, 2 * unused + 2], got_e[:, 0])
    # device path with a sample index that does not exist: its rows are NaN, the others are not
    import torch
    d_in = c.upload_inputs(x)
    bad = torch.tensor([1, 7], dtype=torch.int32, device="cuda")
    e2, m2 = c.psr_full(theta, d_in, sel=bad, obs_list=obs, n_marg_bits=3)
    e2, m2 = e2.cpu().numpy(), m2.cpu().numpy()
    assert np.isnan(e2[1]).all() and np.isnan(m2[1]).all()
    assert np.isfinite(e2[0]).all() and np.isfinite(m2[0]).all()
    assert e2.shape == (2, K, 3)


def _axes_circuit(n, layers):
    """Every qubit gets RX, RZ, RY, RX with parameters of their own in each layer (one fused 2x2 per qubit and
    layer, four parameters in it), between them a CZ ring, a CZ ladder, Z gates and one ZZ sandwich with a
    parameter in its RZ: parameters that resume behind their own layer on register, lane and wave bits, and
    one that patches a phase slot."""
    c = LearningCircuit(n)
    rng = np.random.default_rng(n)
    for q in range(n):
        c.add_input_RY_gate(q, lambda x, q=q: x[q % len(x)])
    for layer in range(layers):
        for q in range(n):
            c.add_parametric_RX_gate(q, rng.uniform(-3, 3))
            c.add_parametric_RZ_gate(q, rng.uniform(-3, 3))
            c.add_parametric_RY_gate(q, rng.uniform(-3, 3))
            c.add_parametric_RX_gate(q, rng.uniform(-3, 3))
        step = 1 if layer % 2 == 0 else 2
        for q in range(n):
            c.add_CZ_gate(q, (q + step) % n)
        c.add_Z_gate((3 * layer + 1) % n)
        c.add_CNOT_gate(layer % n, (layer + 5) % n)
        c.add_parametric_RZ_gate((layer + 5) % n, rng.uniform(-3, 3))
        c.add_CNOT_gate(layer % n, (layer + 5) % n)
    for q in range(n):
        c.add_parametric_RY_gate(q, rng.uniform(-3, 3))
    return c


@pytest.mark.parametrize("n", [10, 12, 13, 14])
def test_sweep_resumes_behind_the_parameters_layer(n, monkeypatch):
    """The sweep kernel of the register engine on a circuit whose parameters sit on every kind of position
    (register, lane and wave bits) with rotations about all three axes: every row against the oracle; the
    same circuit compiled without the search for the initial layout gives the same rows."""
    rng = np.random.default_rng(40 + n)
    x = rng.uniform(0, np.pi, (2, 5))
    o1 = Observable(n); o1.add_operator(1.0, "Z 0"); o1.add_operator(-0.4, f"Z 3 Z {n - 1}")
    o2 = Observable(n); o2.add_operator(0.7, f"Z {n // 2}")
    rows = []
    for search in (True, False):
        if not search:
            monkeypatch.setenv("VQA_REG_NO_LAYOUT_SEARCH", "1")
        c = _axes_circuit(n, 3)
        theta = np.random.default_rng(n).uniform(-3, 3, c.n_theta)
        ia = c.input_angles(x)
        got_e, got_m = c.psr_full(theta, ia, obs_list=[o1, o2], n_marg_bits=2)
        rows.append((got_e, got_m))
        if search:
            si, sp, sv = cpu_oracle.psr_lists(2, c.n_theta)
            want_e, want_m = cpu_oracle.eval_batch(c, theta, ia, si, sp, sv, obs_list=obs_tuples([o1, o2]), n_marg_bits=2)
            assert np.abs(got_e.reshape(want_e.shape) - want_e).max() < TOL
            assert np.abs(got_m.reshape(want_m.shape) - want_m).max() < TOL
    assert np.abs(rows[0][0] - rows[1][0]).max() < TOL and np.abs(rows[0][1] - rows[1][1]).max() < TOL


@pytest.mark.parametrize("n,shared", [(6, False), (7, True), (8, False), (9, True)])
def test_grouped_small_sweeps(n, shared):
    """Below 10 qubits the register engine runs a sweep as groups of shifts of one sample that share
    the sample's prepared tables (each state patches the entries of its own parameter).  Scaled
    parameters, one that drives no gate, one shared by two fused 2x2s; against the oracle and against
    the same kernel family run as an explicit list; parameter subsets; an invalid sample index."""
    import os
    import torch
    set_engine("reg")
    c = _edge_circuit(n, shared)
    rng = np.random.default_rng(n)
    theta = rng.uniform(-3, 3, c.n_theta)
    x = rng.uniform(0, np.pi, (5, 4))
    ia = c.input_angles(x)
    o1 = Observable(n); o1.add_operator(0.6, f"Z 0 Z {n - 1}"); o1.add_operator(0.3, "")
    o2 = Observable(n); o2.add_operator(1.0, f"Z {n // 2}")
    obs = [o1, Observable(n), o2]
    sel = np.array([4, 0, 2], np.int32)
    si, sp, sv = cpu_oracle.psr_lists(3, c.n_theta, sel=sel)
    want_e, want_m = cpu_oracle.eval_batch(c, theta, ia, si, sp, sv, obs_list=obs_tuples(obs), n_marg_bits=3)
    got_e, got_m = c.psr_full(theta, ia, sel=sel, obs_list=obs, n_marg_bits=3)
    assert np.abs(got_e.reshape(want_e.shape) - want_e).max() < TOL
    assert np.abs(got_m.reshape(want_m.shape) - want_m).max() < TOL
    unused = 2 * n
    assert np.array_equal(got_e[:, 2 * unused + 1], got_e[:, 0]) and np.array_equal(got_e[:, 2 * unused + 2], got_e[:, 0])
    os.environ["VQA_NO_GROUP"] = "1"
    try:
        plain_e, plain_m = c.psr_full(theta, ia, sel=sel, obs_list=obs, n_marg_bits=3)
    finally:
        del os.environ["VQA_NO_GROUP"]
    assert np.abs(plain_e - got_e).max() < 1e-13 and np.abs(plain_m - got_m).max() < 1e-13
    # a single-qubit observable: parameters outside its light cone are copied, not run
    z0 = Observable(n); z0.add_operator(1.0, f"Z {n // 2}")
    d_in = c.upload_inputs(x)
    full_e, full_m = c.psr_full(theta, d_in, obs_list=[z0], n_marg_bits=0)
    si5, sp5, sv5 = cpu_oracle.psr_lists(5, c.n_theta)
    want5, _ = cpu_oracle.eval_batch(c, theta, ia, si5, sp5, sv5, obs_list=obs_tuples([z0]))
    assert np.abs(full_e.cpu().numpy().reshape(want5.shape) - want5).max() < TOL
    mine = np.arange(0, c.n_theta, 3)
    e, _ = c.psr_full(theta, d_in, obs_list=[z0], params=mine)
    rows = np.concatenate([[0], 2 * mine + 1, 2 * mine + 2])
    assert float((e[:, rows] - full_e[:, rows]).abs().max()) < TOL
    bad = torch.tensor([1, 9], dtype=torch.int32, device="cuda")
    e2, m2 = c.psr_full(theta, d_in, sel=bad, obs_list=obs, n_marg_bits=3)
    e2, m2 = e2.cpu().numpy(), m2.cpu().numpy()
    assert np.isnan(e2[1]).all() and np.isnan(m2[1]).all()
    assert np.isfinite(e2[0]).all() and np.isfinite(m2[0]).all()


@pytest.mark.parametrize("n", [4, 7, 10, 12])
def test_xy_terms_behind_basis_change(n):
    """Automatic engine, observables with X / Y / mixed Pauli strings: the host layer deals the terms
    into qubit-wise compatible groups and measures each as Z-strings behind H / RZ(pi/2) H on the
    register engine; explicit lists and full sweeps against the oracle, and against the LDS engine's
    direct X/Y measurement."""
    import os
    c = create_ibm_HEA_ansatz(n, 2, seed=5)
    rng = np.random.default_rng(40 + n)
    theta = rng.uniform(0, 2 * np.pi, c.n_theta)
    x = rng.uniform(0, np.pi, (3, n))
    ia = c.input_angles(x)
    h = Observable(n)
    for q in range(n):
        h.add_operator(0.7 - 0.1 * q, f"Z {q} Z {(q + 1) % n}")
        h.add_operator(0.3 + 0.05 * q, f"X {q}")
    o2 = Observable(n); o2.add_operator(0.4, f"X 0 Y {n - 1}"); o2.add_operator(-0.6, "Y 1 Z 2"); o2.add_operator(0.2, "")
    o3 = Observable(n); o3.add_operator(1.0, "Y 0 Y 1 X 2"); o3.add_operator(0.5, "Z 3")
    obs = [h, o2, o3]
    assert c._basis_plan(obs) is not None and len(c._basis_plan(obs)) >= 3
    assert c._basis_plan(obs, 1) is None and c._basis_plan(obs, 1 << 20) is not None   # small calls stay direct
    si, sp, sv = cpu_oracle.psr_lists(3, c.n_theta)
    want_e, want_m = cpu_oracle.eval_batch(c, theta, ia, si, sp, sv, obs_list=obs_tuples(obs), n_marg_bits=2)
    pick = np.arange(0, len(si), 7)
    d_in = c.upload_inputs(x)
    os.environ["VQA_BASIS_CHANGE_ALWAYS"] = "1"
    try:
        got_e, got_m = c.psr_full(theta, ia, obs_list=obs, n_marg_bits=2)
        e2, m2 = c.eval_batch(theta, d_in, si[pick], sp[pick], sv[pick], obs_list=obs, n_marg_bits=2)
    finally:
        del os.environ["VQA_BASIS_CHANGE_ALWAYS"]
    assert np.abs(got_e.reshape(want_e.shape) - want_e).max() < TOL
    assert np.abs(got_m.reshape(want_m.shape) - want_m).max() < TOL
    assert np.abs(e2.cpu().numpy() - want_e[pick]).max() < TOL
    assert np.abs(m2.cpu().numpy() - want_m[pick]).max() < TOL
    os.environ["VQA_NO_BASIS_CHANGE"] = "1"
    try:
        e3, _ = c.eval_batch(theta, d_in, si[pick], sp[pick], sv[pick], obs_list=obs)
    finally:
        del os.environ["VQA_NO_BASIS_CHANGE"]
    assert np.abs(e3.cpu().numpy() - e2.cpu().numpy()).max() < 1e-12


@pytest.mark.parametrize("n", [4, 6, 9, 10, 12])
@pytest.mark.parametrize("n_par", [0, 1, 3])
def test_sweeps_of_tiny_parameter_sets(n, n_par):
    """Sweeps whose run list is empty or shorter than a chunk / a slice: no parameter at all, one
    parameter in a circuit without a diagonal stage, three parameters of which one drives nothing;
    one sample and several."""
    set_engine("reg")
    c = LearningCircuit(n)
    for q in range(n):
        c.add_input_RY_gate(q, lambda x, q=q: x[q % len(x)])
    if n_par >= 1:
        c.add_parametric_RX_gate(n - 1, 0.4)
    if n_par >= 3:
        for q in range(n - 1):
            c.add_CZ_gate(q, q + 1)
        c.add_parametric_RY_gate(0, 0.2)
        c._theta.append(0.1)                       # a parameter that drives no gate
        c.add_H_gate(1)
    rng = np.random.default_rng(n + n_par)
    theta = rng.uniform(-3, 3, c.n_theta)
    zz = Observable(n); zz.add_operator(0.8, f"Z 0 Z {n - 1}"); zz.add_operator(-0.3, "Z 1")
    for B in (1, 5):
        x = rng.uniform(0, np.pi, (B, 2))
        ia = c.input_angles(x)
        si, sp, sv = cpu_oracle.psr_lists(B, c.n_theta)
        want_e, want_m = cpu_oracle.eval_batch(c, theta, ia, si, sp, sv, obs_list=obs_tuples([zz]), n_marg_bits=2)
        got_e, got_m = c.psr_full(theta, ia, obs_list=[zz], n_marg_bits=2)
        assert got_e.shape == (B, 2 * c.n_theta + 1, 1)
        assert np.abs(got_e.reshape(want_e.shape) - want_e).max() < TOL
        assert np.abs(got_m.reshape(want_m.shape) - want_m).max() < TOL


def test_tiled_passes_against_gate_sweeps_22_qubits():
    """Beyond what the oracle finishes quickly: the tiled passes of the HBM engine against its
    one-sweep-per-gate form (independent kernels) on a 22-qubit circuit with all gate kinds."""
    res = []
    for engine in ("hbm", "pergate"):
        set_engine(engine)
        c = random_circuit(LearningCircuit, 22, 120, seed=77)
        rng = np.random.default_rng(8)
        theta = rng.uniform(-3, 3, c.n_theta)
        x = rng.uniform(-1, 1, (2, 3))
        obs = random_observables(22, seed=5, xy=True)
        sp = np.array([-1, 2, 4], np.int32)
        e, m = c.eval_batch(theta, c.input_angles(x), np.array([0, 1, 1], np.int32), sp,
                            np.array([0.0, 0.5 * np.pi, -0.5 * np.pi]), obs_list=obs, n_marg_bits=3)
        res.append((e, m))
    assert np.abs(res[0][0] - res[1][0]).max() < TOL
    assert np.abs(res[0][1] - res[1][1]).max() < TOL
    assert np.abs(res[0][1].sum(axis=1) - 1).max() < TOL
