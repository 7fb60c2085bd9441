"""The C-ABI shared library loads and exports every symbol include/vqa_hip.h
declares; argument errors come back as codes + messages, not crashes.  No
compute call is made here (no GPU needed)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from vqa_amd import _lib
from vqa_amd.circuit import GateList, OP_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vqa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vqa_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported():
    L = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/vqa_hip.h but not exported"
    assert set(syms) == set(_lib.EXPORTS)


def test_version_and_struct_layout():
    assert _lib.lib().vqa_version() == 1
    assert OP_DTYPE.itemsize == 40
    assert [OP_DTYPE.fields[k][1] for k in ("kind", "q0", "q1", "src", "idx", "reserved", "coef", "offset")] == [
        0, 4, 8, 12, 16, 20, 24, 32]


def test_circuit_create_validates_without_gpu():
    L = _lib.lib()
    g = GateList(3)
    g.add_H_gate(0); g.add_parametric_RY_gate(1, 0.3); g.add_CNOT_gate(0, 2)
    ops = g.op_table()
    h = C.c_void_p()
    assert L.vqa_circuit_create(3, _lib.nptr(ops), len(ops), 1, 0, C.byref(h)) == 0
    n = [C.c_int() for _ in range(4)]
    assert L.vqa_circuit_info(h, *[C.byref(v) for v in n]) == 0
    assert [v.value for v in n] == [3, 3, 1, 0]
    assert L.vqa_circuit_destroy(h) == 0
    # bad qubit index
    bad = ops.copy(); bad[0]["q0"] = 7
    assert L.vqa_circuit_create(3, _lib.nptr(bad), len(bad), 1, 0, C.byref(h)) == _lib.EINVAL
    assert b"out of range" in L.vqa_last_error()
    # theta index beyond n_theta
    assert L.vqa_circuit_create(3, _lib.nptr(ops), len(ops), 0, 0, C.byref(h)) == _lib.EINVAL
    # unknown gate kind
    bad = ops.copy(); bad[1]["kind"] = 99
    assert L.vqa_circuit_create(3, _lib.nptr(bad), len(bad), 1, 0, C.byref(h)) == _lib.EINVAL
    assert L.vqa_circuit_create(0, None, 0, 0, 0, C.byref(h)) == _lib.EINVAL
    assert L.vqa_circuit_destroy(None) == 0


def test_circuit_engine_choice_without_gpu():
    """The engine that serves Z-basis calls, by register size (automatic choice): LDS below four
    qubits, registers up to fourteen, HBM tiles beyond; NULL handle is an error."""
    L = _lib.lib()
    assert L.vqa_set_engine(0) == 0
    for n, want in ((3, 1), (4, 2), (9, 2), (12, 2), (14, 2), (15, 3), (22, 3)):
        g = GateList(n)
        for q in range(n):
            g.add_H_gate(q); g.add_parametric_RY_gate(q, 0.1 * q)
        for q in range(n - 1):
            g.add_CZ_gate(q, q + 1)
        ops = g.op_table()
        h = C.c_void_p()
        assert L.vqa_circuit_create(n, _lib.nptr(ops), len(ops), n, 0, C.byref(h)) == 0
        assert L.vqa_circuit_engine(h) == want, n
        assert L.vqa_circuit_destroy(h) == 0
    assert L.vqa_circuit_engine(None) == _lib.EINVAL


def test_observable_create_validates():
    L = _lib.lib()
    h = C.c_void_p()
    coefs = np.array([1.0, 0.5]); zm = np.array([1, 3], np.uint64)
    assert L.vqa_observable_create(2, _lib.nptr(coefs), _lib.nptr(zm), None, 2, C.byref(h)) == 0
    assert L.vqa_observable_destroy(h) == 0
    zm = np.array([1, 4], np.uint64)
    assert L.vqa_observable_create(2, _lib.nptr(coefs), _lib.nptr(zm), None, 2, C.byref(h)) == _lib.EINVAL
    with pytest.raises(_lib.VqaError):
        _lib.check(L.vqa_set_engine(9))
