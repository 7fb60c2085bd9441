"""Pauli-sum target object on the CPU-oracle circuit (host logic), and driven by the
reference's VQE optimizers when that package is importable."""
import numpy as np
import pytest

from oracle_circuit import OracleCircuit
from vqa_amd.observable import Observable
from vqa_amd.qml_utils import create_ry_cz_ansatz
from vqa_amd.vqe_target import PauliSumVQE


def tfim(n, J=1.0, h=1.0, const=0.25):
    H = Observable(n)
    for i in range(n):
        H.add_operator(J, f"Z {i} Z {(i + 1) % n}")
        H.add_operator(h, f"X {i}")
    H.add_operator(const, "")
    return H


def make(n=5, d=2, min_approx=0, cls=OracleCircuit):
    c = create_ry_cz_ansatz(n, d, seed=3, circuit_cls=cls)
    v = PauliSumVQE(n, c, tfim(n), min_approx_shots=min_approx)
    rng = np.random.default_rng(1)
    return v, rng.uniform(-np.pi, np.pi, v.n_params)


def dense_energy(v, params):
    from oracle import cpu_oracle
    psi = cpu_oracle.statevector(v.circuit, params, v.circuit.input_row(np.zeros(1)))
    n = v.n_qubits
    X = np.array([[0, 1], [1, 0]], complex); Z = np.diag([1.0, -1.0]).astype(complex); I = np.eye(2, dtype=complex)

    def op(paulis):
        m = np.array([[1.0 + 0j]])
        for q in reversed(range(n)):
            m = np.kron(m, {"X": X, "Z": Z}.get(paulis.get(q), I))
        return m

    e = 0.0
    for t in v.hamiltonian.terms:
        e += t.coef * np.real(np.vdot(psi, op(t.paulis) @ psi))
    return e


def test_exact_energy_and_gradient():
    v, p = make(min_approx=-10)
    assert abs(v.exact_energy(p) - dense_energy(v, p)) < 1e-10
    out = v.iEval(p, np.full(v.n_params, 100))
    h = 1e-6
    for i in (0, 3, v.n_params - 1):
        e = np.zeros_like(p); e[i] = h
        fd = (v.exact_energy(p + e) - v.exact_energy(p - e)) / (2 * h)
        assert abs(fd - out["grad"][i]) < 1e-7
    assert v.shots_counter[0] == 0 and np.all(out["gvar_list"] >= 0)
    assert np.allclose(v.grad_eval(p, np.full(v.n_params, 100)), out["grad"])


def test_shot_noise_statistics_and_counters():
    v, p = make(min_approx=0)
    exact = PauliSumVQE(5, v.circuit, tfim(5), min_approx_shots=-10).iEval(p, np.full(v.n_params, 1))
    np.random.seed(0)
    n = 40000
    reps = np.array([v.iEval(p, np.full(v.n_params, n))["grad"] for _ in range(30)])
    sigma = np.sqrt(exact["gvar_list"] / n)
    assert np.all(np.abs(reps.mean(0) - exact["grad"]) < 5 * sigma / np.sqrt(30) + 1e-4)
    # the quoted one-shot variance predicts the spread (floor() leaves a few shots unused)
    assert np.all(np.abs(reps.std(0, ddof=1) / sigma - 1.0) < 0.45)
    v.shots_counter[0] = v.circuit_counter[0] = 0
    v.energy_eval(p, 1000)
    assert v.circuit_counter[0] == 1 and 1000 - len(v.terms) <= v.shots_counter[0] <= 1000 + len(v.terms)


def test_reference_vqe_optimizer_drives_the_target(tmp_path, monkeypatch):
    ref = pytest.importorskip("vqa.VQA_optimizers.vqe_optimizers")
    monkeypatch.chdir(tmp_path)   # the reference writes log files into the working directory
    v, p = make(n=4, d=2, min_approx=0)
    np.random.seed(5)
    opt = ref.Adam(hpara=dict(LR_rule=lambda t: 0.1, s_list_rule=lambda t: np.full(v.n_params, 400),
                              LR_annealing_by_shots=False, s_list_annealing_by_shots=False),
                   init_spec_dict={"target_obj": v, "init_params": p.copy()})
    res = opt.optimize(options=dict(max_iter=25, track_var_names=[], track_indicator="iteration_num",
                                    take_SA=False, take_BMA=False))
    assert res["func_evol"][-1] < res["func_evol"][0] - 0.5
    assert res["total_shots"][-1] > 0
