"""Gate-list IR and the LearningCircuit stand-in that feeds the HIP engine.

The reference builds its circuits with skqulacs' ``LearningCircuit`` and runs
them one sample at a time (``circuit.update_parameters(theta)`` then
``circuit.run(x)``; vqa/QNN/myQNNs_base.py:1110-1124).  This module keeps that
builder / duck-typed interface (the methods the reference calls:
vqa/QNN/QML_utils.py:47-98, 114-163; vqa/QNN/myQNNs_base.py:290, 745, 758, 913)
and adds what the batched engine needs: a flat op table (``vqa_op`` records of
include/vqa_hip.h) and a per-dataset matrix of input-gate angles, because the
Python ``input_func`` lambdas cannot run on the device.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np

# mirrors `struct vqa_op` in include/vqa_hip.h (40 bytes, natural alignment)
OP_DTYPE = np.dtype(
    [
        ("kind", "<i4"),
        ("q0", "<i4"),
        ("q1", "<i4"),
        ("src", "<i4"),
        ("idx", "<i4"),
        ("reserved", "<i4"),
        ("coef", "<f8"),
        ("offset", "<f8"),
    ],
    align=True,
)
assert OP_DTYPE.itemsize == 40

GATE_H, GATE_X, GATE_Y, GATE_Z, GATE_RX, GATE_RY, GATE_RZ, GATE_CNOT, GATE_CZ = range(9)
SRC_CONST, SRC_THETA, SRC_INPUT = range(3)

_ROT = {"X": GATE_RX, "Y": GATE_RY, "Z": GATE_RZ}


@dataclass(frozen=True)
class GateSpec:
    """A fixed gate handed to ``LearningCircuit.add_gate`` (stands in for a
    qulacs gate object such as ``qulacs.gate.CZ(i, j)``, vqa/QNN/QML_utils.py:91)."""

    kind: int
    q0: int
    q1: int = -1
    angle: float = 0.0


def H(i: int) -> GateSpec:
    return GateSpec(GATE_H, i)


def X(i: int) -> GateSpec:
    return GateSpec(GATE_X, i)


def Y(i: int) -> GateSpec:
    return GateSpec(GATE_Y, i)


def Z(i: int) -> GateSpec:
    return GateSpec(GATE_Z, i)


def CNOT(control: int, target: int) -> GateSpec:
    return GateSpec(GATE_CNOT, target, control)


def CZ(i: int, j: int) -> GateSpec:
    return GateSpec(GATE_CZ, i, j)


def RX(i: int, angle: float) -> GateSpec:
    return GateSpec(GATE_RX, i, -1, angle)


def RY(i: int, angle: float) -> GateSpec:
    return GateSpec(GATE_RY, i, -1, angle)


def RZ(i: int, angle: float) -> GateSpec:
    return GateSpec(GATE_RZ, i, -1, angle)


class GateList:
    """Host-side gate list: the builder half of a LearningCircuit.

    Holds the submitted, unfused gates in order, the current values of the
    trainable parameters and the Python ``input_func`` of every input gate.
    """

    def __init__(self, n_qubit: int):
        if not (1 <= int(n_qubit) <= 30):
            raise ValueError("n_qubit must be in [1, 30]")
        self.n_qubit = int(n_qubit)
        self._ops: List[tuple] = []
        self._theta: List[float] = []
        self._input_funcs: List[Callable] = []

    # ---- fixed gates -------------------------------------------------
    def _check(self, *qs: int) -> None:
        for q in qs:
            if not (0 <= int(q) < self.n_qubit):
                raise ValueError(f"qubit index {q} out of range for {self.n_qubit} qubits")

    def _push(self, kind, q0, q1=-1, src=SRC_CONST, idx=0, coef=0.0, offset=0.0):
        self._check(q0)
        if q1 >= 0:
            self._check(q1)
            if q1 == q0:
                raise ValueError("two-qubit gate needs two different qubits")
        self._ops.append((kind, int(q0), int(q1), src, int(idx), 0, float(coef), float(offset)))

    def add_gate(self, gate: GateSpec) -> None:
        self._push(gate.kind, gate.q0, gate.q1, SRC_CONST, 0, 0.0, gate.angle)

    def add_H_gate(self, index: int) -> None:
        self._push(GATE_H, index)

    def add_X_gate(self, index: int) -> None:
        self._push(GATE_X, index)

    def add_Y_gate(self, index: int) -> None:
        self._push(GATE_Y, index)

    def add_Z_gate(self, index: int) -> None:
        self._push(GATE_Z, index)

    def add_CNOT_gate(self, control_index: int, target_index: int) -> None:
        self._push(GATE_CNOT, target_index, control_index)

    def add_CZ_gate(self, index0: int, index1: int) -> None:
        self._push(GATE_CZ, index0, index1)

    def add_RX_gate(self, index: int, angle: float) -> None:
        self._push(GATE_RX, index, offset=angle)

    def add_RY_gate(self, index: int, angle: float) -> None:
        self._push(GATE_RY, index, offset=angle)

    def add_RZ_gate(self, index: int, angle: float) -> None:
        self._push(GATE_RZ, index, offset=angle)

    # ---- input (data-encoding) gates ------------------------------------
    def _add_input(self, axis: str, index: int, input_func: Callable) -> None:
        self._input_funcs.append(input_func)
        self._push(_ROT[axis], index, src=SRC_INPUT, idx=len(self._input_funcs) - 1, coef=1.0)

    def add_input_RX_gate(self, index: int, input_func: Callable = lambda x: x[0]) -> None:
        self._add_input("X", index, input_func)

    def add_input_RY_gate(self, index: int, input_func: Callable = lambda x: x[0]) -> None:
        self._add_input("Y", index, input_func)

    def add_input_RZ_gate(self, index: int, input_func: Callable = lambda x: x[0]) -> None:
        self._add_input("Z", index, input_func)

    # ---- trainable gates ----------------------------------------------
    def _add_parametric(self, axis, index, parameter, share_with=None, share_with_coef=None) -> int:
        if share_with is None:
            self._theta.append(float(parameter))
            pid, coef = len(self._theta) - 1, 1.0
        else:
            if not (0 <= share_with < len(self._theta)):
                raise ValueError("share_with refers to a parameter that does not exist")
            pid, coef = int(share_with), 1.0 if share_with_coef is None else float(share_with_coef)
        self._push(_ROT[axis], index, src=SRC_THETA, idx=pid, coef=coef)
        return pid

    def add_parametric_RX_gate(self, index, parameter, share_with=None, share_with_coef=None) -> int:
        return self._add_parametric("X", index, parameter, share_with, share_with_coef)

    def add_parametric_RY_gate(self, index, parameter, share_with=None, share_with_coef=None) -> int:
        return self._add_parametric("Y", index, parameter, share_with, share_with_coef)

    def add_parametric_RZ_gate(self, index, parameter, share_with=None, share_with_coef=None) -> int:
        return self._add_parametric("Z", index, parameter, share_with, share_with_coef)

    # ---- parameters -----------------------------------------------------
    def get_parameters(self) -> List[float]:
        return list(self._theta)

    def update_parameters(self, theta: Sequence[float]) -> None:
        theta = [float(t) for t in theta]
        if len(theta) != len(self._theta):
            raise ValueError(f"expected {len(self._theta)} parameters, got {len(theta)}")
        self._theta = theta

    # ---- tables for the engine -----------------------------------------
    @property
    def n_theta(self) -> int:
        return len(self._theta)

    @property
    def n_input(self) -> int:
        return len(self._input_funcs)

    @property
    def n_ops(self) -> int:
        return len(self._ops)

    def op_table(self) -> np.ndarray:
        """The flat ``vqa_op`` array handed to ``vqa_circuit_create``."""
        tab = np.zeros(len(self._ops), dtype=OP_DTYPE)
        for i, op in enumerate(self._ops):
            tab[i] = op
        return tab

    def input_angles(self, x_list) -> np.ndarray:
        """``[B, n_input]`` matrix of input-gate angles: row b holds
        ``input_func(x_b)`` of every input gate, in gate order."""
        x_arr = np.asarray(x_list, dtype=float)
        if x_arr.ndim == 1:
            x_arr = x_arr[:, None] if self.n_input == 0 or x_arr.size == 0 else x_arr.reshape(len(x_arr), -1)
        out = np.empty((len(x_arr), self.n_input), dtype=np.float64)
        for b, x in enumerate(x_arr):
            for g, f in enumerate(self._input_funcs):
                out[b, g] = f(x)
        return np.ascontiguousarray(out)

    def input_row(self, x) -> np.ndarray:
        x = np.atleast_1d(np.asarray(x, dtype=float))
        return np.array([f(x) for f in self._input_funcs], dtype=np.float64)
