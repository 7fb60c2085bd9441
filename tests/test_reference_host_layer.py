"""The QNN target objects against what the reference's OWN classes return
(tests/golden/host_layer.json, recorded by tests/golden/make_host_layer.py from
vqa/QNN/myQNNs_base.py, myQNNs_classifier.py and myQNNs_regression.py running on stand-ins for
their absent imports; see that script's header for what the stand-ins are).

Every recorded call -- exact forward functions, the parameter-shift gradient oracles in exact,
CLT, mixed and finite-shot modes, loss estimators, variance estimators, the gCANS-style random
allocation, the classifier's MSE / cross-entropy oracles -- is replayed in order on a fresh
target object of this package with the same seed; results and counter increments must agree.
On CPU the circuit is the oracle circuit; under -m gpu it is the HIP circuit.  What this pins is
the reference's Python algebra (shot allocation, estimators, generator order, counters); the
statevector arithmetic underneath is the oracle's on both sides."""
import json
import os

import numpy as np
import pytest

from oracle_circuit import OracleCircuit
from vqa_amd import qml_utils
from vqa_amd.qnn_base import myQNN_base, myQNN_regression
from vqa_amd.qnn_classifier import myQNN_classifier

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "host_layer.json")) as f:
    GOLD = json.load(f)
CASES = GOLD["cases"]
IDS = [f"{c['cls']}-{c['builder']}-{c['n_qubit']}q" for c in CASES]
CLASSES = {"myQNN_base": myQNN_base, "myQNN_regression": myQNN_regression, "myQNN_classifier": myQNN_classifier}
TOL = 2e-9   # gradients are means of products of O(1) numbers: rounding differences stay far below this


def dec(v):
    if isinstance(v, dict):
        if "nd" in v:
            return np.asarray(v["nd"])
        if "list" in v:
            return [dec(x) for x in v["list"]]
        if "dict" in v:
            return {k: dec(x) for k, x in v["dict"].items()}
    return v


def close(got, want, path):
    if isinstance(want, dict):
        assert isinstance(got, dict) and set(want) <= set(got), path
        for k in want:
            close(got[k], want[k], f"{path}.{k}")
    elif isinstance(want, list):
        assert len(got) == len(want), path
        for i, (g, w) in enumerate(zip(got, want)):
            close(g, w, f"{path}[{i}]")
    elif want is None:
        assert got is None, path
    else:
        g = np.asarray(got.detach().cpu().numpy() if hasattr(got, "detach") else got, dtype=float)
        w = np.asarray(want, dtype=float)
        assert g.shape == w.shape, f"{path}: shape {g.shape} != {w.shape}"
        err = np.abs(g - w) / np.maximum(1.0, np.abs(w))
        assert err.size == 0 or err.max() < TOL, f"{path}: max deviation {err.max():.3e}"


def replay(case, circuit_cls):
    circuit = getattr(qml_utils, case["builder"])(case["n_qubit"], case["depth"], case["builder_seed"], circuit_cls=circuit_cls)
    qnn = CLASSES[case["cls"]](case["n_qubit"], circuit, np.asarray(case["x_train"]), np.asarray(case["y_train"]),
                               np.asarray(case["x_test"]), np.asarray(case["y_test"]), shot_draws="per_shot",
                               file_id="fixture", **case["ctor"])
    for k, call in enumerate(case["calls"]):
        np.random.seed(call["seed"])
        before = [qnn.shots_counter[0], qnn.call_counter[0], qnn.circuit_counter[0], qnn.epoch_counter[0]]
        out = getattr(qnn, call["method"])(*[dec(a) for a in call["args"]], **{n: dec(v) for n, v in call["kwargs"].items()})
        where = f"call {k} {call['method']}({call['kwargs']})"
        close(out, dec(call["out"]), where)
        after = [qnn.shots_counter[0], qnn.call_counter[0], qnn.circuit_counter[0], qnn.epoch_counter[0]]
        assert [int(a - b) for a, b in zip(after, before)] == call["counters"], where + ": counters"


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_replay_on_oracle_circuit(case):
    replay(case, OracleCircuit)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_replay_on_gpu(case):
    from vqa_amd import LearningCircuit

    replay(case, LearningCircuit)
