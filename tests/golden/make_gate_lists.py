"""Generates gate_lists.json: the gate lists the REFERENCE's own ansatz builders produce.

    cd /tmp && python /root/repo/tests/golden/make_gate_lists.py

Runs /root/reference/vqa/QNN/QML_utils.py (create_jerbi_ansatz, create_ibm_HEA_ansatz; pure
Python) in the build container with RECORDING stand-ins for the two imports it cannot
resolve here -- skqulacs.circuit.LearningCircuit and qulacs.gate.CZ, neither installed.  The
stand-ins compute nothing: they write down every builder call (gate kind, qubits, initial
angle) and keep the input_func closures, which are then evaluated on seeded feature vectors
exactly as LearningCircuit.run(x) would evaluate them (after the builder has returned, so
late-bound closure variables have their final values).

What the vectors pin: gate order, qubit pairs, control/target roles, the initial-angle draws
and every input angle of both reference ansaetze.  Only data is written (no reference text).
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/vqa/QNN/QML_utils.py"

CASES = [  # (builder, n_qubit, train_depth, seed, n_features)
    ("create_jerbi_ansatz", 2, 0, 0, 2),
    ("create_jerbi_ansatz", 4, 0, 3, 4),
    ("create_jerbi_ansatz", 4, 1, 0, 4),
    ("create_jerbi_ansatz", 5, 2, 11, 5),
    ("create_jerbi_ansatz", 6, 3, 5, 2),
    ("create_jerbi_ansatz", 3, 1, 2, 1),
    ("create_ibm_HEA_ansatz", 3, 1, 0, 1),
    ("create_ibm_HEA_ansatz", 2, 0, 4, 2),
    ("create_ibm_HEA_ansatz", 4, 1, 0, 4),
    ("create_ibm_HEA_ansatz", 5, 2, 9, 5),
    ("create_ibm_HEA_ansatz", 6, 5, 0, 1),
    ("create_ibm_HEA_ansatz", 7, 3, 1, 3),
]
N_X = 3  # feature vectors per case


class RecordingCircuit:
    """Stands in for skqulacs.circuit.LearningCircuit: records, computes nothing."""

    def __init__(self, n_qubit):
        self.n_qubit = n_qubit
        self.gates = []        # [kind, q0, q1, source, index]
        self.theta0 = []
        self.input_funcs = []

    def add_H_gate(self, index):
        self.gates.append(["H", int(index), -1, "const", 0])

    def add_CNOT_gate(self, control_index, target_index):
        self.gates.append(["CNOT", int(target_index), int(control_index), "const", 0])

    def add_gate(self, gate):
        self.gates.append(list(gate))

    def add_input_RZ_gate(self, index, input_func=lambda x: x[0]):
        self.input_funcs.append(input_func)
        self.gates.append(["RZ", int(index), -1, "input", len(self.input_funcs) - 1])

    def _parametric(self, axis, index, parameter):
        self.theta0.append(float(parameter))
        self.gates.append(["R" + axis, int(index), -1, "theta", len(self.theta0) - 1])

    def add_parametric_RX_gate(self, index, parameter):
        self._parametric("X", index, parameter)

    def add_parametric_RY_gate(self, index, parameter):
        self._parametric("Y", index, parameter)

    def add_parametric_RZ_gate(self, index, parameter):
        self._parametric("Z", index, parameter)


def load_reference_builders():
    skq, skq_c = types.ModuleType("skqulacs"), types.ModuleType("skqulacs.circuit")
    skq_c.LearningCircuit = RecordingCircuit
    qul, qul_g = types.ModuleType("qulacs"), types.ModuleType("qulacs.gate")
    qul_g.CZ = lambda i, j: ("CZ", int(i), int(j), "const", 0)
    stand_ins = {"skqulacs": skq, "skqulacs.circuit": skq_c, "qulacs": qul, "qulacs.gate": qul_g}
    saved = {k: sys.modules.get(k) for k in stand_ins}
    sys.modules.update(stand_ins)
    # the reference annotates with np.float_, an alias NumPy 2 dropped (annotation only)
    if not hasattr(np, "float_"):
        np.float_ = np.float64
    try:
        spec = importlib.util.spec_from_file_location("_reference_QML_utils", REF)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def main():
    ref = load_reference_builders()
    out = []
    for builder, n, depth, seed, n_feat in CASES:
        circuit = getattr(ref, builder)(n, depth, seed)
        rng = np.random.default_rng(1000 + 17 * n + depth)
        xs = rng.uniform(0.1, 3.0, (N_X, n_feat))
        angles = [[float(f(x)) for f in circuit.input_funcs] for x in xs]
        out.append(dict(builder=builder, n_qubit=n, train_depth=depth, seed=seed, gates=circuit.gates,
                        theta0=circuit.theta0, x=xs.tolist(), input_angles=angles))
        print(builder, n, depth, seed, len(circuit.gates), "gates", len(circuit.theta0), "parameters",
              len(circuit.input_funcs), "input gates")
    path = os.path.join(HERE, "gate_lists.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
