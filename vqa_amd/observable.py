"""Observable stand-in for ``qulacs.Observable`` as the reference uses it.

The reference builds observables with ``Observable(n)`` + ``add_operator(coef,
"Z 0 Z 1")`` and reads ``get_expectation_value(state)`` (vqa/QNN/myQNNs_base.py:
353-361, 692-704; vqa/QNN/myQNNs_classifier.py:295-340), plus ``to_json``,
``get_term_count`` and ``get_term(i).get_index_list()`` at construction
(vqa/QNN/myQNNs_base.py:325-330).  Here an observable is a list of
(coef, zmask, xmask) Pauli strings that the HIP engine reduces on the device.
"""
from __future__ import annotations

import ctypes as C
import json
from typing import List

import numpy as np

from . import _lib


class PauliTerm:
    def __init__(self, coef: float, paulis: dict):
        self.coef = float(coef)
        self.paulis = dict(paulis)  # qubit -> 'X' | 'Y' | 'Z'

    def get_index_list(self) -> List[int]:
        return sorted(self.paulis)

    def get_pauli_id_list(self) -> List[int]:
        return [{"X": 1, "Y": 2, "Z": 3}[self.paulis[q]] for q in sorted(self.paulis)]

    def get_coef(self) -> complex:
        return complex(self.coef)

    @property
    def zmask(self) -> int:
        return sum(1 << q for q, p in self.paulis.items() if p in "ZY")

    @property
    def xmask(self) -> int:
        return sum(1 << q for q, p in self.paulis.items() if p in "XY")


class Observable:
    def __init__(self, n_qubits: int):
        self.n_qubits = int(n_qubits)
        self.terms: List[PauliTerm] = []
        self._handle = None

    # ---- qulacs-style builder ---------------------------------------------
    def add_operator(self, coef, pauli_string: str = "") -> None:
        if isinstance(coef, complex):
            if abs(coef.imag) > 0:
                raise ValueError("only real coefficients are supported")
            coef = coef.real
        toks = pauli_string.split()
        if len(toks) % 2:
            raise ValueError(f"malformed Pauli string {pauli_string!r}")
        paulis = {}
        for p, q in zip(toks[0::2], toks[1::2]):
            p, q = p.upper(), int(q)
            if p not in ("X", "Y", "Z", "I"):
                raise ValueError(f"unknown Pauli {p!r}")
            if not (0 <= q < self.n_qubits):
                raise ValueError(f"qubit {q} out of range")
            if q in paulis:
                raise ValueError(f"qubit {q} appears twice in {pauli_string!r}")
            if p != "I":
                paulis[q] = p
        self.terms.append(PauliTerm(coef, paulis))
        self._release()

    def get_term_count(self) -> int:
        return len(self.terms)

    def get_term(self, i: int) -> PauliTerm:
        return self.terms[i]

    def get_qubit_count(self) -> int:
        return self.n_qubits

    def to_json(self) -> str:
        return json.dumps(
            {
                "name": "Observable",
                "qubit_count": self.n_qubits,
                "operators": [
                    {
                        "coef": {"real": t.coef, "imag": 0.0},
                        "pauli_string": " ".join(f"{t.paulis[q]} {q}" for q in sorted(t.paulis)),
                    }
                    for t in self.terms
                ],
            }
        )

    # ---- arrays / handle for the engine ----------------------------------------
    def arrays(self):
        coefs = np.array([t.coef for t in self.terms], np.float64)
        zm = np.array([t.zmask for t in self.terms], np.uint64)
        xm = np.array([t.xmask for t in self.terms], np.uint64)
        return coefs, zm, xm

    def handle(self):
        if self._handle is None:
            coefs, zm, xm = self.arrays()
            h = C.c_void_p()
            _lib.check(
                _lib.lib().vqa_observable_create(
                    self.n_qubits, _lib.nptr(coefs), _lib.nptr(zm), _lib.nptr(xm), len(coefs), C.byref(h)
                )
            )
            self._handle = h
        return self._handle

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            _lib.lib().vqa_observable_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_handle"] = None
        return d

    # ---- per-state expectation (plumbing shim) -----------------------------
    def get_expectation_value(self, state) -> complex:
        """Expectation on a state returned by ``LearningCircuit.run``; evaluated
        by the HIP engine as a one-instance batch."""
        return complex(state._expectation(self))


def z_observable(n_qubits: int, qubit: int = 0) -> Observable:
    """The regressor's observable, Z on one qubit (vqa/QNN/myQNNs_base.py:353-361)."""
    obs = Observable(n_qubits)
    obs.add_operator(1.0, f"Z {qubit}")
    return obs
