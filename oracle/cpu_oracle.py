"""ctypes binding of oracle/liboracle.so (oracle.c).

TEST INFRASTRUCTURE ONLY -- the checker the HIP engine is compared with, and
the reported CPU baseline.  Parity unpinned: see the header of oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    path = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return path


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        dp, ip, up = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)
        L.oracle_eval_batch.restype = C.c_int
        L.oracle_eval_batch.argtypes = [
            C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
            dp, dp, C.c_int, ip, ip, dp, C.c_int64,
            dp, up, up, ip, C.c_int, C.c_int, dp, dp, C.c_int,
        ]
        L.oracle_statevector.restype = C.c_int
        L.oracle_statevector.argtypes = [C.c_int, C.c_void_p, C.c_int, dp, dp, C.c_int, C.c_double, dp]
        _LIB = L
    return _LIB


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def flatten_obs(obs_list):
    """obs_list: sequence of (coefs, zmasks[, xmasks]) -> flat arrays + offsets."""
    coefs, zm, xm, off = [], [], [], [0]
    for o in obs_list:
        c, z = o[0], o[1]
        x = o[2] if len(o) > 2 and o[2] is not None else [0] * len(c)
        coefs += list(c); zm += list(z); xm += list(x)
        off.append(len(coefs))
    return (np.asarray(coefs, np.float64), np.asarray(zm, np.uint64),
            np.asarray(xm, np.uint64), np.asarray(off, np.int32))


def eval_batch(circuit, theta, input_angles, sample_idx=None, shift_param=None, shift_val=None,
               obs_list=(), n_marg_bits=0, n_threads=0):
    """CPU counterpart of vqa_eval_batch for a GateList-like `circuit`.
    Returns (expect[E, n_obs], marginal[E, 2^m] or None)."""
    ops = circuit.op_table()
    theta = np.ascontiguousarray(theta, np.float64)
    ia = None if input_angles is None else np.ascontiguousarray(input_angles, np.float64)
    B = 0 if ia is None else ia.shape[0]
    if sample_idx is not None:
        sample_idx = np.ascontiguousarray(sample_idx, np.int32)
        E = len(sample_idx)
    elif shift_param is not None:
        E = len(shift_param)
    else:
        E = max(B, 1)
    if shift_param is not None:
        shift_param = np.ascontiguousarray(shift_param, np.int32)
        shift_val = np.ascontiguousarray(shift_val, np.float64)
    coefs, zm, xm, off = flatten_obs(obs_list)
    n_obs = len(obs_list)
    out_e = np.zeros((E, max(n_obs, 1)), np.float64)
    out_m = np.zeros((E, 1 << n_marg_bits), np.float64) if n_marg_bits > 0 else None
    rc = lib().oracle_eval_batch(
        circuit.n_qubit, ops.ctypes.data_as(C.c_void_p), len(ops), circuit.n_theta, circuit.n_input,
        _p(theta, C.c_double), _p(ia, C.c_double), B, _p(sample_idx, C.c_int32),
        _p(shift_param, C.c_int32), _p(shift_val, C.c_double), E,
        _p(coefs, C.c_double), _p(zm, C.c_uint64), _p(xm, C.c_uint64), _p(off, C.c_int32),
        n_obs, n_marg_bits, _p(out_e, C.c_double), _p(out_m, C.c_double), n_threads)
    if rc != 0:
        raise RuntimeError(f"oracle_eval_batch failed with code {rc}")
    return out_e[:, :n_obs], out_m


def psr_lists(n_sel: int, P: int, shift: float = 0.5 * np.pi, sel: Optional[Sequence[int]] = None):
    """Instance list of a full parameter-shift sweep, in vqa_psr_full's order."""
    K = 2 * P + 1
    s = np.arange(n_sel, dtype=np.int32) if sel is None else np.asarray(sel, np.int32)
    sample_idx = np.repeat(s, K)
    sp = np.full(K, -1, np.int32)
    sv = np.zeros(K)
    sp[1::2] = np.arange(P); sp[2::2] = np.arange(P)
    sv[1::2] = shift; sv[2::2] = -shift
    return sample_idx, np.tile(sp, n_sel), np.tile(sv, n_sel)


def statevector(circuit, theta, input_row=None, shift_param=-1, shift_val=0.0):
    ops = circuit.op_table()
    theta = np.ascontiguousarray(theta, np.float64)
    row = None if input_row is None else np.ascontiguousarray(input_row, np.float64)
    out = np.zeros(2 << circuit.n_qubit, np.float64)
    rc = lib().oracle_statevector(circuit.n_qubit, ops.ctypes.data_as(C.c_void_p), len(ops),
                                  _p(theta, C.c_double), _p(row, C.c_double),
                                  int(shift_param), float(shift_val), _p(out, C.c_double))
    if rc != 0:
        raise RuntimeError(f"oracle_statevector failed with code {rc}")
    return out.view(np.complex128)
