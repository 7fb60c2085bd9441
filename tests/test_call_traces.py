"""Replay of the calls the reference's SantaQlaus / gCANS / SPSA optimizers made into this
package's regressor (tests/golden/call_traces.json, recorded with the reference imported in
the build container by tests/golden/make_call_traces.py): same oracle, same positional and
keyword arguments, same order, NumPy re-seeded per call as at recording time -- the answers
and what each call adds to the shot / call / circuit / epoch counters must come back.  On the CPU-oracle circuit
(host logic) and, marked gpu, on the HIP circuit through the C ABI."""
import json
import os

import numpy as np
import pytest

from oracle_circuit import OracleCircuit
from vqa_amd.qml_utils import create_ibm_HEA_ansatz
from vqa_amd.qnn_base import myQNN_base

HERE = os.path.dirname(os.path.abspath(__file__))


def dec(v):
    if isinstance(v, dict):
        if "nd" in v:
            return np.array(v["nd"], dtype=v["dtype"])
        if "dict" in v:
            return {k: dec(x) for k, x in v["dict"].items()}
        if "list" in v:
            return [dec(x) for x in v["list"]]
    return v


def replay(name, circuit_cls, tol):
    with open(os.path.join(HERE, "golden", "trajectories.json")) as f:
        ds = json.load(f)["dataset"]
    with open(os.path.join(HERE, "golden", "call_traces.json")) as f:
        traces = json.load(f)
    circuit = create_ibm_HEA_ansatz(ds["n_qubit"], ds["depth"], circuit_cls=circuit_cls)
    qnn = myQNN_base(ds["n_qubit"], circuit, np.array(ds["x_train"]), np.array(ds["y_train"]), np.array(ds["x_test"]),
                     np.array(ds["y_test"]), min_approx_shots=0)
    qnn.set_methods_for_optimizer(myQNN_base._LOSS_METHODS["mean_squared_error"])
    calls = traces["cases"][name]
    assert len(calls) >= 13
    for i, c in enumerate(calls):
        np.random.seed(traces["seed0"] + i)
        before = [qnn.shots_counter[0], qnn.call_counter[0], qnn.circuit_counter[0], qnn.epoch_counter[0]]
        got = qnn.methods_for_optimization[c["method"]](*[dec(a) for a in c["args"]],
                                                        **{k: dec(v) for k, v in c["kwargs"].items()})
        want = dec(c["out"])
        if isinstance(want, dict):
            for k, w in want.items():
                assert np.max(np.abs(np.asarray(got[k], dtype=float) - np.asarray(w, dtype=float))) < tol, (name, i, c["method"], k)
        else:
            assert abs(float(got) - float(want)) < tol, (name, i, c["method"])
        after = [qnn.shots_counter[0], qnn.call_counter[0], qnn.circuit_counter[0], qnn.epoch_counter[0]]
        assert [a - b for a, b in zip(after, before)] == c["counters"], (name, i, c["method"])


@pytest.mark.parametrize("name", ["SantaQlaus", "gCANS", "SPSA"])
def test_reference_call_traces_oracle_circuit(name):
    replay(name, OracleCircuit, 1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["SantaQlaus", "gCANS", "SPSA"])
def test_reference_call_traces_hip_circuit(name):
    from vqa_amd import LearningCircuit

    replay(name, LearningCircuit, 1e-9)
