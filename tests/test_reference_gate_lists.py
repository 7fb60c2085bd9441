"""The ansatz builders against the gate lists the reference's own builders produce
(tests/golden/gate_lists.json, written by tests/golden/make_gate_lists.py from
vqa/QNN/QML_utils.py:33-164 run with recording stand-ins for LearningCircuit / CZ): gate kinds,
order, qubits, control/target roles, initial angles, and every input angle on seeded feature
vectors.  On the GPU the recorded gate lists themselves (rebuilt from the fixture rows, not
through this package's builders) run through the HIP path and are compared with the oracle."""
import json
import os

import numpy as np
import pytest

from helpers import TOL
from oracle import cpu_oracle
from vqa_amd import circuit as vc
from vqa_amd import qml_utils
from vqa_amd.circuit import GateList
from vqa_amd.observable import Observable

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "gate_lists.json")) as f:
    CASES = json.load(f)
IDS = [f"{c['builder']}-{c['n_qubit']}q-d{c['train_depth']}-s{c['seed']}" for c in CASES]

KIND = {"H": vc.GATE_H, "CNOT": vc.GATE_CNOT, "CZ": vc.GATE_CZ, "RX": vc.GATE_RX, "RY": vc.GATE_RY, "RZ": vc.GATE_RZ}
SRC = {"const": vc.SRC_CONST, "theta": vc.SRC_THETA, "input": vc.SRC_INPUT}


def canonical(gl):
    """(kind, q0, q1, src, idx) rows of a GateList; CZ pairs as the reference passed them."""
    return [(int(o[0]), int(o[1]), int(o[2]), int(o[3]) if o[0] in (vc.GATE_RX, vc.GATE_RY, vc.GATE_RZ) else 0,
             int(o[4])) for o in gl._ops]


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_builder_matches_reference_gate_list(case):
    gl = getattr(qml_utils, case["builder"])(case["n_qubit"], case["train_depth"], case["seed"], circuit_cls=GateList)
    want = [(KIND[k], q0, q1, SRC[s] if k in ("RX", "RY", "RZ") else 0, idx) for k, q0, q1, s, idx in case["gates"]]
    assert canonical(gl) == want
    assert gl.get_parameters() == case["theta0"]          # bit-identical draws
    got = gl.input_angles(np.asarray(case["x"]))
    assert got.shape == np.asarray(case["input_angles"]).shape
    assert np.array_equal(got, np.asarray(case["input_angles"]))


def from_fixture(cls, case):
    """A circuit built row by row from the recorded gate list; its input gates read column g of
    a feature vector that IS the recorded input-angle row."""
    c = cls(case["n_qubit"])
    for k, q0, q1, s, idx in case["gates"]:
        if k == "H":
            c.add_H_gate(q0)
        elif k == "CNOT":
            c.add_CNOT_gate(q1, q0)
        elif k == "CZ":
            c.add_CZ_gate(q0, q1)
        elif s == "theta":
            getattr(c, f"add_parametric_{k}_gate")(q0, case["theta0"][idx])
        elif s == "input":
            getattr(c, f"add_input_{k}_gate")(q0, lambda x, g=idx: x[g])
        else:
            raise AssertionError(k)
    return c


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_recorded_gate_list_equals_builder_on_oracle(case):
    """Oracle values of the recorded list and of this package's builder agree (they are the same
    gates: a cross-check that the fixture rows are read with the right roles)."""
    rec = from_fixture(GateList, case)
    gl = getattr(qml_utils, case["builder"])(case["n_qubit"], case["train_depth"], case["seed"], circuit_cls=GateList)
    n = case["n_qubit"]
    obs = [Observable(n)]
    obs[0].add_operator(1.0, "Z 0")
    theta = np.asarray(case["theta0"])
    ia = np.asarray(case["input_angles"])
    e1, m1 = cpu_oracle.eval_batch(rec, theta, ia, obs_list=[o.arrays() for o in obs], n_marg_bits=min(2, n))
    e2, m2 = cpu_oracle.eval_batch(gl, theta, gl.input_angles(np.asarray(case["x"])),
                                   obs_list=[o.arrays() for o in obs], n_marg_bits=min(2, n))
    assert np.array_equal(e1, e2) and np.array_equal(m1, m2)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_recorded_gate_list_on_gpu(case):
    from vqa_amd import LearningCircuit

    n = case["n_qubit"]
    hip = from_fixture(LearningCircuit, case)
    ora = from_fixture(GateList, case)
    obs = [Observable(n), Observable(n)]
    obs[0].add_operator(1.0, "Z 0")
    obs[1].add_operator(0.5, f"Z 0 Z {n - 1}" if n > 1 else "Z 0")
    obs[1].add_operator(0.25, "")
    theta = np.asarray(case["theta0"])
    ia = np.asarray(case["input_angles"])
    mb = min(2, n)
    # full parameter-shift sweeps of the recorded samples, every instance against the oracle
    si, sp, sv = cpu_oracle.psr_lists(len(ia), len(theta))
    we, wm = cpu_oracle.eval_batch(ora, theta, ia, si, sp, sv, [o.arrays() for o in obs], mb)
    ge, gm = hip.psr_full(theta, ia, obs_list=obs, n_marg_bits=mb)
    assert np.abs(np.asarray(ge).reshape(we.shape) - we).max() < TOL
    assert np.abs(np.asarray(gm).reshape(wm.shape) - wm).max() < TOL
    ge, gm = hip.eval_batch(theta, ia, si, sp, sv, obs_list=obs, n_marg_bits=mb)
    assert np.abs(np.asarray(ge) - we).max() < TOL
    assert np.abs(np.asarray(gm) - wm).max() < TOL
