"""The light-cone report of the C ABI (vqa_dead_parameters; host-only) against the oracle: a
parameter reported dead must leave every observable and the marginal unchanged under its
parameter shift -- on the reference's ansaetze and on random circuits with every gate kind."""
import numpy as np
import pytest

from vqa_amd import LearningCircuit, Observable
from vqa_amd.qml_utils import create_ibm_HEA_ansatz, create_jerbi_ansatz, create_ry_cz_ansatz
from oracle import cpu_oracle
from helpers import obs_tuples, random_circuit, random_observables


def check(c, obs, marg, x, theta, expect_some_dead=None):
    dead = c.dead_parameters(obs, marg)
    assert dead.shape == (c.n_theta,)
    ia = c.input_angles(x)
    si, sp, sv = cpu_oracle.psr_lists(len(x), c.n_theta)
    e, m = cpu_oracle.eval_batch(c, theta, ia, si, sp, sv, obs_list=obs_tuples(obs), n_marg_bits=marg)
    K = 2 * c.n_theta + 1
    e = e.reshape(len(x), K, -1)
    dev = np.abs(e[:, 1:, :] - e[:, :1, :]).max(axis=(0, 2)).reshape(c.n_theta, 2).max(axis=1)
    if marg:
        m = m.reshape(len(x), K, -1)
        dev = np.maximum(dev, np.abs(m[:, 1:, :] - m[:, :1, :]).max(axis=(0, 2)).reshape(c.n_theta, 2).max(axis=1))
    assert dev[dead].max(initial=0.0) < 1e-12, "a parameter reported dead changes the measured values"
    if expect_some_dead is not None:
        assert dead.sum() == expect_some_dead
    return dead, dev


def test_ibm_hea_z0_readout():
    """Depth-2 IBM-HEA on 8 qubits, <Z0> and the two-bit marginal: the last layer only matters on the
    readout qubits (and not through its trailing RZ), the layer before on their neighbours, ..."""
    c = create_ibm_HEA_ansatz(8, 2, seed=0, circuit_cls=LearningCircuit)
    rng = np.random.default_rng(0)
    z0 = Observable(8); z0.add_operator(1.0, "Z 0")
    dead, dev = check(c, [z0], 2, rng.uniform(0, np.pi, (2, 8)), rng.uniform(0, 2 * np.pi, c.n_theta))
    P = c.n_theta
    last = np.arange(P - 24, P).reshape(8, 3)           # (RX, RY, RZ) of qubit q in the last layer
    assert dead[last[2:]].all() and not dead[last[:2, :2]].any() and dead[last[:2, 2]].all()
    # last layer: qubits 2..7 and two trailing RZ; the one before: qubits 3..6 + two RZ; the first: qubits 4, 5 + two RZ
    assert dead.sum() == (18 + 2) + (12 + 2) + (6 + 2)
    assert (dev[~dead] > 1e-9).mean() > 0.9             # the live ones do matter here
    # a global observable keeps every rotation but the trailing RZs alive
    zall = Observable(8); zall.add_operator(1.0, " ".join(f"Z {q}" for q in range(8)))
    dead2, _ = check(c, [zall], 0, rng.uniform(0, np.pi, (1, 8)), rng.uniform(0, 2 * np.pi, P))
    assert dead2.sum() == 8 and dead2[last[:, 2]].all()


def test_benchmark_circuit_count():
    c = create_ibm_HEA_ansatz(12, 10, seed=0, circuit_cls=LearningCircuit)
    z0 = Observable(12); z0.add_operator(1.0, "Z 0")
    assert c.dead_parameters([z0], 0).sum() >= c.dead_parameters([z0], 2).sum() == 102


@pytest.mark.parametrize("seed", range(6))
def test_random_circuits(seed):
    n = 4 + seed % 4
    c = random_circuit(LearningCircuit, n, 50, seed=900 + seed)
    rng = np.random.default_rng(seed)
    obs = random_observables(n, seed=seed, xy=bool(seed % 2))[:2]
    local = Observable(n); local.add_operator(0.7, f"Z {n - 1}")
    check(c, obs + [local], seed % 3, rng.uniform(-1, 1, (2, 3)), rng.uniform(-3, 3, c.n_theta))
    check(c, [local], 0, rng.uniform(-1, 1, (2, 3)), rng.uniform(-3, 3, c.n_theta))


@pytest.mark.parametrize("builder", [create_jerbi_ansatz, create_ry_cz_ansatz])
def test_other_ansaetze(builder):
    c = builder(6, 2, seed=1, circuit_cls=LearningCircuit)
    rng = np.random.default_rng(3)
    o = Observable(6); o.add_operator(1.0, "Z 2"); o.add_operator(0.5, "X 3")
    check(c, [o], 1, rng.uniform(0, np.pi, (2, 6)), rng.uniform(-3, 3, c.n_theta))


# ---- light-cone groups (host-side planning, no GPU) ---------------------------------------------
def _ising_terms(n):
    from vqa_amd import Observable

    out = []
    for i in range(n):
        for s in (f"Z {i} Z {(i + 1) % n}", f"X {i}"):
            o = Observable(n)
            o.add_operator(1.0, s)
            out.append(o)
    return out


def test_light_cone_groups_of_the_ising_ring():
    """28 qubits, RY layers with six CZ rings between them: the cone of Z_i Z_{i+1} is 14 qubits wide, that of
    X_i 13; the ZZ terms run on their own, the X terms two to a group (their basis differs from the ZZ terms'
    on the shared qubit): 42 groups of 14 qubits.  A forced engine keeps the whole register."""
    from vqa_amd import LearningCircuit, _lib
    from vqa_amd.qml_utils import create_ry_cz_ansatz

    n = 28
    c = create_ry_cz_ansatz(n, 6, seed=4, circuit_cls=LearningCircuit)
    g = c.light_cone_groups(_ising_terms(n))
    assert len(g) == 42 and set(g.tolist()) == {14}
    # depth 2: cones of 5 and 6 qubits; a handful of wider groups is cheaper than 56 launches
    c2 = create_ry_cz_ansatz(n, 2, seed=4, circuit_cls=LearningCircuit)
    g2 = c2.light_cone_groups(_ising_terms(n))
    assert 0 < len(g2) < 42 and g2.max() <= 14
    # many samples: the work outweighs the launches, narrower groups
    g3 = c2.light_cone_groups(_ising_terms(n), n_instances=4096)
    assert len(g3) >= len(g2) and g3.max() <= g2.max()
    _lib.check(_lib.lib().vqa_set_engine(3))
    try:
        c3 = create_ry_cz_ansatz(n, 6, seed=4, circuit_cls=LearningCircuit)
        assert len(c3.light_cone_groups(_ising_terms(n))) == 0
    finally:
        _lib.lib().vqa_set_engine(0)


def test_light_cone_groups_keep_a_full_cone_whole():
    """The 20-qubit IBM-HEA circuit of depth 8 with <Z0>: behind eight CZ rings and the two ring feature maps
    the cone is the whole register -- nothing to split.  At depth 1 the cone is 7 qubits wide."""
    from vqa_amd import LearningCircuit, Observable
    from vqa_amd.qml_utils import create_ibm_HEA_ansatz

    z0 = Observable(20)
    z0.add_operator(1.0, "Z 0")
    deep = create_ibm_HEA_ansatz(20, 8, seed=0, circuit_cls=LearningCircuit)
    assert len(deep.light_cone_groups([z0], n_instances=1024)) == 0
    assert len(deep.light_cone_groups([z0], n_marg_bits=2, n_instances=1024)) == 0
    shallow = create_ibm_HEA_ansatz(20, 1, seed=0, circuit_cls=LearningCircuit)
    g = shallow.light_cone_groups([z0], n_instances=64)
    assert g.tolist() == [7]
