"""Second, structurally independent oracle: dense Kronecker-product matrices in
NumPy (n <= 10).  TEST INFRASTRUCTURE ONLY.  Shares no code with oracle.c: it
builds each gate as a full 2^n x 2^n matrix, so an indexing mistake in the
pair-sweep code of oracle.c (or of the HIP kernels) cannot hide in both."""
from __future__ import annotations

import numpy as np

I2 = np.eye(2, dtype=complex)
PX = np.array([[0, 1], [1, 0]], dtype=complex)
PY = np.array([[0, -1j], [1j, 0]], dtype=complex)
PZ = np.array([[1, 0], [0, -1]], dtype=complex)
HAD = np.array([[1, 1], [1, -1]], dtype=complex) / np.sqrt(2)
P0 = np.array([[1, 0], [0, 0]], dtype=complex)
P1 = np.array([[0, 0], [0, 1]], dtype=complex)


def embed(n, mats):
    """kron with qubit 0 as the least significant index bit: mats = {qubit: 2x2}."""
    out = np.array([[1.0 + 0j]])
    for q in range(n - 1, -1, -1):
        out = np.kron(out, mats.get(q, I2))
    return out


def rot(P, a):
    # qulacs: R_P(a) = exp(+i a/2 P) = cos(a/2) I + i sin(a/2) P
    return np.cos(a / 2) * I2 + 1j * np.sin(a / 2) * P


def gate_matrix(n, kind, q0, q1, angle):
    if kind == 0: return embed(n, {q0: HAD})
    if kind == 1: return embed(n, {q0: PX})
    if kind == 2: return embed(n, {q0: PY})
    if kind == 3: return embed(n, {q0: PZ})
    if kind == 4: return embed(n, {q0: rot(PX, angle)})
    if kind == 5: return embed(n, {q0: rot(PY, angle)})
    if kind == 6: return embed(n, {q0: rot(PZ, angle)})
    if kind == 7:  # CNOT: q1 control, q0 target
        return embed(n, {q1: P0}) + embed(n, {q1: P1, q0: PX})
    if kind == 8:
        return embed(n, {}) - 2 * embed(n, {q0: P1, q1: P1})
    raise ValueError(kind)


def statevector(circuit, theta, input_row=None, shift_param=-1, shift_val=0.0):
    n = circuit.n_qubit
    assert n <= 10
    psi = np.zeros(1 << n, dtype=complex); psi[0] = 1
    for op in circuit.op_table():
        v = 0.0
        if op["src"] == 1:
            v = theta[op["idx"]] + (shift_val if op["idx"] == shift_param else 0.0)
        elif op["src"] == 2:
            v = input_row[op["idx"]]
        angle = op["coef"] * v + op["offset"]
        psi = gate_matrix(n, int(op["kind"]), int(op["q0"]), int(op["q1"]), angle) @ psi
    return psi


def pauli_matrix(n, zmask, xmask=0):
    mats = {}
    for q in range(n):
        z, x = (zmask >> q) & 1, (xmask >> q) & 1
        if z and x: mats[q] = PY
        elif z: mats[q] = PZ
        elif x: mats[q] = PX
    return embed(n, mats)


def expectation(psi, n, coefs, zmasks, xmasks=None):
    tot = 0.0
    for t, c in enumerate(coefs):
        Pm = pauli_matrix(n, int(zmasks[t]), int(xmasks[t]) if xmasks is not None else 0)
        tot += c * np.real(np.vdot(psi, Pm @ psi))
    return tot


def marginal(psi, m):
    p = np.abs(psi) ** 2
    return p.reshape(-1, 1 << m).sum(axis=0)
