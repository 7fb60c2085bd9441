"""Ansatz builders: gate lists of the circuits the reference trains.

Same gate order, qubit pairing, input functions and initial-angle draws as
vqa/QNN/QML_utils.py:33-164, expressed on this package's LearningCircuit
(a gate-list builder) instead of skqulacs'.  ``create_ry_rz_cz_ansatz`` and
``create_ry_cz_ansatz`` are the two benchmark ansaetze that have no builder in
the reference (4-qubit Iris classifier, 28-qubit Pauli-sum circuit).
"""
from __future__ import annotations

import numpy as np
from numpy.random import default_rng


def _default_circuit_cls():
    from .learning_circuit import LearningCircuit

    return LearningCircuit


def _preprocess_x(x, index: int) -> float:
    # vqa/QNN/QML_utils.py:43-45, 110-112: features are reused cyclically
    return x[index % len(x)]


def _add_hea_block(circuit, n_qubit: int, train_depth: int, seed: int) -> None:
    """Trainable block shared by both reference ansaetze
    (vqa/QNN/QML_utils.py:80-98, 145-163): per layer RX, RY, RZ on every qubit
    with angles 2*pi*rng.random() drawn in gate order, then a CZ ring; one more
    rotation layer at the end."""
    rng = default_rng(seed)
    for _ in range(train_depth):
        for i in range(n_qubit):
            circuit.add_parametric_RX_gate(i, 2.0 * np.pi * rng.random())
            circuit.add_parametric_RY_gate(i, 2.0 * np.pi * rng.random())
            circuit.add_parametric_RZ_gate(i, 2.0 * np.pi * rng.random())
        for i in range(n_qubit):
            circuit.add_CZ_gate(i, (i + 1) % n_qubit)
    for i in range(n_qubit):
        circuit.add_parametric_RX_gate(i, 2.0 * np.pi * rng.random())
        circuit.add_parametric_RY_gate(i, 2.0 * np.pi * rng.random())
        circuit.add_parametric_RZ_gate(i, 2.0 * np.pi * rng.random())


def create_jerbi_ansatz(n_qubit: int, train_depth: int, seed: int = 0, circuit_cls=None,
                        reference_closure: bool = True):
    """All-pairs ZZ feature map (a CNOT-RZ-CNOT sandwich for every j < i),
    applied twice, followed by the trainable block.  vqa/QNN/QML_utils.py:33-99.

    ``reference_closure``: the reference's ZZ lambdas bind ``i`` as a default
    argument but read ``j`` from the enclosing scope (QML_utils.py:58-60,
    74-76), and ``j`` is ONE function-local cell that the CZ loop of the
    trainable block assigns as well (``j = (i + 1) % n_qubit``, QML_utils.py:90).
    When ``run(x)`` evaluates the lambdas ``j`` is whatever was assigned last:
    0 for ``train_depth >= 1`` (the CZ ring ends on ``(n-1+1) % n``), and
    ``n_qubit - 2`` for ``train_depth == 0`` (the feature-map loop ends there).
    Every ZZ angle is therefore ``x_i * x_0`` (``x_i * x_{n-2}`` without
    trainable layers).  True (default) reproduces exactly that; False gives
    the per-pair product ``x_i * x_j``."""
    cls = circuit_cls or _default_circuit_cls()
    circuit = cls(n_qubit)
    j_seen = 0 if train_depth >= 1 else n_qubit - 2
    for _rep in range(2):
        for i in range(n_qubit):
            circuit.add_H_gate(i)
        for i in range(n_qubit):
            circuit.add_input_RZ_gate(i, lambda x, i=i: _preprocess_x(x, i))
            for j in range(i):
                circuit.add_CNOT_gate(i, j)
                jj = j_seen if reference_closure else j
                circuit.add_input_RZ_gate(
                    j, lambda x, i=i, jj=jj: _preprocess_x(x, i) * _preprocess_x(x, jj)
                )
                circuit.add_CNOT_gate(i, j)
    _add_hea_block(circuit, n_qubit, train_depth, seed)
    return circuit


def create_ibm_HEA_ansatz(n_qubit: int, train_depth: int, seed: int = 0, circuit_cls=None,
                          reference_closure: bool = True):
    """IBM-style ZZ feature map on ring neighbours, applied twice, followed by
    the trainable block.  vqa/QNN/QML_utils.py:101-164.

    ``reference_closure``: as in ``create_jerbi_ansatz`` the reference's ZZ
    lambdas read ``j`` late (QML_utils.py:124-126, 139-141); the loops end on
    ``j = 0``, so every ZZ angle is ``(pi - x_i) * (pi - x_0)``.  True (default)
    reproduces that; False gives ``(pi - x_i) * (pi - x_{i+1})``."""
    cls = circuit_cls or _default_circuit_cls()
    circuit = cls(n_qubit)
    for _rep in range(2):
        for i in range(n_qubit):
            circuit.add_H_gate(i)
        for i in range(n_qubit):
            j = (i + 1) % n_qubit
            circuit.add_input_RZ_gate(i, lambda x, i=i: _preprocess_x(x, i))
            circuit.add_CNOT_gate(i, j)
            jj = 0 if reference_closure else j
            circuit.add_input_RZ_gate(
                j,
                lambda x, i=i, jj=jj: (np.pi - _preprocess_x(x, i)) * (np.pi - _preprocess_x(x, jj)),
            )
            circuit.add_CNOT_gate(i, j)
    _add_hea_block(circuit, n_qubit, train_depth, seed)
    return circuit


def create_ry_rz_cz_ansatz(n_qubit: int, train_depth: int, seed: int = 0, circuit_cls=None):
    """RY(x_q) encoding on qubit q, then ``train_depth`` layers of (RY, RZ per
    qubit; CZ ring) and a final (RY, RZ) layer; initial angles uniform in
    [0, 2 pi).  The 4-qubit Iris classifier circuit of the benchmark list."""
    cls = circuit_cls or _default_circuit_cls()
    circuit = cls(n_qubit)
    for q in range(n_qubit):
        circuit.add_input_RY_gate(q, lambda x, q=q: _preprocess_x(x, q))
    rng = default_rng(seed)
    for layer in range(train_depth + 1):
        for q in range(n_qubit):
            circuit.add_parametric_RY_gate(q, 2.0 * np.pi * rng.random())
            circuit.add_parametric_RZ_gate(q, 2.0 * np.pi * rng.random())
        if layer < train_depth and n_qubit > 1:
            for q in range(n_qubit):
                circuit.add_CZ_gate(q, (q + 1) % n_qubit)
    return circuit


def create_ry_cz_ansatz(n_qubit: int, train_depth: int, seed: int = 0, circuit_cls=None):
    """``train_depth + 1`` layers of RY on every qubit with a CZ ring between
    consecutive layers; angles uniform in [-pi, pi).  The input-free circuit of
    the Pauli-sum (VQE-style) benchmark."""
    cls = circuit_cls or _default_circuit_cls()
    circuit = cls(n_qubit)
    rng = default_rng(seed)
    for layer in range(train_depth + 1):
        for q in range(n_qubit):
            circuit.add_parametric_RY_gate(q, np.pi * (2.0 * rng.random() - 1.0))
        if layer < train_depth and n_qubit > 1:
            for q in range(n_qubit):
                circuit.add_CZ_gate(q, (q + 1) % n_qubit)
    return circuit
