"""LearningCircuit: the reference's duck-typed circuit object, backed by the
HIP engine.

Drop-in for what the reference asks of ``skqulacs.circuit.LearningCircuit``
(builder methods vqa/QNN/QML_utils.py:47-98; ``get_parameters`` /
``update_parameters`` / ``run`` vqa/QNN/myQNNs_base.py:290, 758, 913) plus the
batched entry points the QNN host layer uses instead of the per-sample loop
(vqa/QNN/myQNNs_base.py:1110-1124): ``eval_batch`` and ``psr_full`` run whole
instance lists on the GPU through the C ABI (include/vqa_hip.h).

Device memory is handled with torch tensors (``device='cuda'``); the compute is
the library's.  Without the built library or without a GPU every device call
raises: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .circuit import GateList
from .observable import Observable


def _torch():
    import torch

    return torch


class State:
    """What ``LearningCircuit.run(x)`` returns: enough of ``qulacs.QuantumState``
    for the reference's per-sample code (``sampling``; expectation through
    ``Observable.get_expectation_value``).  Plumbing shim, not the hot path."""

    def __init__(self, circuit: "LearningCircuit", theta: np.ndarray, row: np.ndarray):
        self._circuit, self._theta, self._row = circuit, theta, row
        self._vec = None

    def get_vector(self) -> np.ndarray:
        if self._vec is None:
            self._vec = self._circuit.statevector(self._theta, self._row)
        return self._vec

    def get_qubit_count(self) -> int:
        return self._circuit.n_qubit

    def sampling(self, n_shots: int, seed: Optional[int] = None):
        """n_shots basis indices drawn from |amp|^2 (NumPy global RNG unless a
        seed is given; qulacs' private stream cannot be reproduced)."""
        p = np.abs(self.get_vector()) ** 2
        p = p / p.sum()
        rng = np.random if seed is None else np.random.default_rng(seed)
        return list(rng.choice(len(p), size=int(n_shots), p=p))

    def _expectation(self, obs: Observable) -> float:
        e, _ = self._circuit.eval_batch(
            self._theta, self._row[None, :], obs_list=[obs], n_marg_bits=0, device_out=False
        )
        return float(e[0, 0])


class LearningCircuit(GateList):
    def __init__(self, n_qubit: int, device: int = 0):
        super().__init__(n_qubit)
        self.device = int(device)
        self._handle = None
        self._handle_nops = -1

    # ---- native handle ---------------------------------------------------
    def _push(self, *a, **k):
        super()._push(*a, **k)
        self._release()

    def handle(self):
        if self._handle is None or self._handle_nops != self.n_ops:
            self._release()
            ops = self.op_table()
            h = C.c_void_p()
            _lib.check(
                _lib.lib().vqa_circuit_create(
                    self.n_qubit, _lib.nptr(ops), len(ops), self.n_theta, self.n_input, C.byref(h)
                )
            )
            self._handle, self._handle_nops = h, self.n_ops
        return self._handle

    def _release(self):
        if getattr(self, "_handle", None) is not None:
            _lib.lib().vqa_circuit_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_handle"] = None
        d["_handle_nops"] = -1
        d.pop("_basis_cache", None)
        return d

    # ---- reference per-sample interface ------------------------------------
    def run(self, x) -> State:
        return State(self, np.asarray(self._theta, np.float64), self.input_row(x))

    def statevector(self, theta, input_row, shift_param: int = -1, shift_val: float = 0.0) -> np.ndarray:
        theta = np.ascontiguousarray(theta, np.float64)
        row = np.ascontiguousarray(input_row, np.float64)
        out = np.empty(2 << self.n_qubit, np.float64)
        _lib.check(
            _lib.lib().vqa_statevector(
                self.handle(), self.device, _lib.nptr(theta), _lib.nptr(row) if row.size else None,
                int(shift_param), float(shift_val), _lib.nptr(out),
            )
        )
        return out.view(np.complex128)

    # ---- batched interface ---------------------------------------------------
    def to_device(self, arr, dtype=None):
        torch = _torch()
        if isinstance(arr, torch.Tensor):
            t = arr.to(device=f"cuda:{self.device}")
            if dtype is not None:
                t = t.to({np.float64: torch.float64, np.int32: torch.int32}[dtype])
            return t.contiguous()
        a = np.ascontiguousarray(arr, dtype=dtype)
        return torch.from_numpy(a).to(f"cuda:{self.device}")

    def upload_inputs(self, x_list):
        """Evaluate every input_func on every sample once and keep the
        ``[B, n_input]`` angle matrix resident on the GPU."""
        return self.to_device(self.input_angles(x_list), np.float64)

    def _on_stream(self, stream):
        """Context in which torch's current stream is `stream` (a torch.cuda.Stream, or a raw
        hipStream_t as an int): the temporaries and outputs of a call are then allocated on the
        stream the library's kernels run on, so the caching allocator cannot hand their memory
        out again while those kernels are still running."""
        torch = _torch()
        if not isinstance(stream, torch.cuda.Stream):
            stream = torch.cuda.ExternalStream(int(stream), device=self.device)
        return torch.cuda.stream(stream)

    def _obs_array(self, obs_list):
        arr = (C.c_void_p * max(len(obs_list), 1))()
        for i, o in enumerate(obs_list):
            arr[i] = o.handle().value
        return arr

    # ---- X / Y Pauli terms behind basis-change gates ---------------------------
    def _basis_plan(self, obs_list, instances=None):
        """The register engine measures in the Z basis.  When it serves this circuit and the
        observables hold X / Y factors, their terms are dealt into groups that agree qubit by
        qubit on the measurement basis; each group is measured as Z-strings on a copy of the
        circuit with H (X) or RZ(pi/2), H (Y) appended to the rotated qubits:
        <psi|Y|psi> = <G psi|Z|G psi> for G = H S^dagger, and RZ(pi/2) is S^dagger up to a phase
        in the convention RZ(a) = diag(e^{ia/2}, e^{-ia/2}).  Group 0 is this circuit itself (pure
        Z terms, and the marginal).  Returns None when nothing is to be split, else a list of
        (circuit, observables), one Z-only observable per caller's observable in every group."""
        if not any(t.xmask for o in obs_list for t in o.terms) or os.environ.get("VQA_NO_BASIS_CHANGE"):
            return None
        # a call or two more per batch: below about 2^20 amplitudes of work (a single small circuit)
        # the direct X/Y measurement of the LDS / streaming engine is a few tenths of a millisecond faster
        # (VQA_BASIS_CHANGE_ALWAYS: tests)
        if instances is not None and (int(instances) << self.n_qubit) < (1 << 20) and not os.environ.get("VQA_BASIS_CHANGE_ALWAYS"):
            return None
        if _lib.lib().vqa_circuit_engine(self.handle()) != 2:
            return None
        groups = [{}]   # qubit -> 'X' | 'Y' | 'Z'
        members = [[]]  # (observable index, coef, support mask)
        for oi, o in enumerate(obs_list):
            for tm in o.terms:
                g = 0
                if tm.xmask:
                    for g in range(1, len(groups) + 1):
                        if g == len(groups):
                            groups.append({})
                            members.append([])
                        if all(groups[g].get(q, p) == p for q, p in tm.paulis.items()):
                            break
                    groups[g].update(tm.paulis)
                members[g].append((oi, tm.coef, tm.zmask | tm.xmask))
        cache = self.__dict__.setdefault("_basis_cache", {})
        if cache.get("n_ops") != self.n_ops:
            cache.clear()
            cache["n_ops"] = self.n_ops
        plan = []
        for g, (basis, mem) in enumerate(zip(groups, members)):
            if g == 0:
                circ = self
            else:
                key = tuple(sorted((q, p) for q, p in basis.items() if p != "Z"))
                circ = cache.get(key)
                if circ is None:
                    circ = LearningCircuit(self.n_qubit, self.device)
                    circ._ops = list(self._ops)
                    circ._theta = list(self._theta)
                    circ._input_funcs = list(self._input_funcs)
                    for q, p in key:
                        if p == "Y":
                            circ.add_RZ_gate(q, 0.5 * np.pi)
                        circ.add_H_gate(q)
                    cache[key] = circ
            obs = [Observable(self.n_qubit) for _ in obs_list]
            for oi, coef, mask in mem:
                obs[oi].add_operator(coef, " ".join(f"Z {q}" for q in range(self.n_qubit) if (mask >> q) & 1))
            plan.append((circ, obs))
        return plan

    def _run_split(self, plan, method, args, kw, n_marg_bits):
        e_tot, m_out = None, None
        for g, (circ, obs) in enumerate(plan):
            k = dict(kw)
            if g > 0:
                k.pop("out_expect", None)
                k.pop("out_marginal", None)
            e, m = getattr(circ, method)(*args, obs_list=obs, n_marg_bits=n_marg_bits if g == 0 else 0, **k)
            if g == 0:
                e_tot, m_out = e, m
            else:
                e_tot += e
        return e_tot, m_out

    def eval_batch(self, theta, inputs, sample_idx=None, shift_param=None, shift_val=None,
                   obs_list: Sequence[Observable] = (), n_marg_bits: int = 0, device_out: bool = False,
                   stream=None):
        """Evaluate an explicit instance list.  ``inputs``: ``[B, n_input]``
        angles (numpy: host-pointer path; torch cuda tensor: device path).
        Returns ``(expect[E, n_obs], marginal[E, 2^m] or None)``."""
        if stream is not None:
            with self._on_stream(stream):
                return self.eval_batch(theta, inputs, sample_idx, shift_param, shift_val, obs_list, n_marg_bits, device_out)
        if sample_idx is not None:
            n_inst = len(sample_idx)
        elif shift_param is not None:
            n_inst = len(shift_param)
        else:
            n_inst = 1 if inputs is None else max(int(inputs.shape[0]), 1)
        plan = self._basis_plan(obs_list, n_inst)
        if plan is not None:
            return self._run_split(plan, "eval_batch", (theta, inputs, sample_idx, shift_param, shift_val),
                                   dict(device_out=device_out), n_marg_bits)
        torch = _torch()
        L = _lib.lib()
        n_obs, M = len(obs_list), 1 << n_marg_bits
        oarr = self._obs_array(obs_list)
        on_dev = isinstance(inputs, torch.Tensor) or device_out
        B = 0 if inputs is None else int(inputs.shape[0])
        if sample_idx is not None:
            E = len(sample_idx)
        elif shift_param is not None:
            E = len(shift_param)
        else:
            E = max(B, 1)
        if not on_dev:
            theta = np.ascontiguousarray(theta, np.float64)
            ia = None if inputs is None else np.ascontiguousarray(inputs, np.float64)
            si = None if sample_idx is None else np.ascontiguousarray(sample_idx, np.int32)
            sp = None if shift_param is None else np.ascontiguousarray(shift_param, np.int32)
            sv = None if shift_val is None else np.ascontiguousarray(shift_val, np.float64)
            oe = np.zeros((E, n_obs), np.float64)
            om = np.zeros((E, M), np.float64) if n_marg_bits > 0 else None
            _lib.check(L.vqa_eval_batch(self.handle(), self.device, _lib.nptr(theta), _lib.nptr(ia), B,
                                        _lib.nptr(si), _lib.nptr(sp), _lib.nptr(sv), E, oarr, n_obs,
                                        n_marg_bits, _lib.nptr(oe) if n_obs else None, _lib.nptr(om)))
            return oe, om
        dev = f"cuda:{self.device}"
        th = self.to_device(theta, np.float64)
        ia = None if inputs is None else self.to_device(inputs, np.float64)
        si = None if sample_idx is None else self.to_device(sample_idx, np.int32)
        sp = None if shift_param is None else self.to_device(shift_param, np.int32)
        sv = None if shift_val is None else self.to_device(shift_val, np.float64)
        oe = torch.empty((E, n_obs), dtype=torch.float64, device=dev)
        om = torch.empty((E, M), dtype=torch.float64, device=dev) if n_marg_bits > 0 else None
        st = torch.cuda.current_stream(self.device).cuda_stream
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        _lib.check(L.vqa_eval_batch_dev(self.handle(), self.device, C.c_void_p(st), p(th), p(ia), B,
                                        p(si), p(sp), p(sv), E, oarr, n_obs, n_marg_bits,
                                        p(oe) if n_obs else None, p(om)))
        return oe, om

    def dead_parameters(self, obs_list: Sequence[Observable] = (), n_marg_bits: int = 0) -> np.ndarray:
        """Mask of the parameters whose rotations lie outside the backward light cone of the
        measured qubits: a sweep returns the unshifted values for them without running the circuit."""
        out = np.zeros(max(self.n_theta, 1), np.uint8)
        _lib.check(_lib.lib().vqa_dead_parameters(self.handle(), self._obs_array(obs_list), len(obs_list), n_marg_bits,
                                                  _lib.nptr(out)))
        return out[:self.n_theta].astype(bool)

    def light_cone_groups(self, obs_list: Sequence[Observable] = (), n_marg_bits: int = 0, sweep: bool = True,
                          n_instances: int = 1) -> np.ndarray:
        """Register sizes of the light-cone groups the library would deal the terms of ``obs_list`` into for a
        sweep of ``n_instances`` samples (``sweep=False``: an explicit list of that many evaluations); an empty
        array when the call runs on the whole register.  Host only."""
        out = np.zeros(4096, np.int32)
        n = _lib.check(_lib.lib().vqa_light_cone_groups(self.handle(), self._obs_array(obs_list), len(obs_list), n_marg_bits,
                                                        int(sweep), int(n_instances), _lib.nptr(out), len(out)))
        return out[:min(n, len(out))].copy()

    def psr_full(self, theta, inputs, sel=None, n_sel: Optional[int] = None,
                 shift: float = 0.5 * np.pi, obs_list: Sequence[Observable] = (), n_marg_bits: int = 0,
                 stream=None, out_expect=None, out_marginal=None, params=None):
        """Full parameter-shift sweep (2P+1 instances per selected sample).
        numpy ``inputs``: host-pointer path, returns numpy; torch cuda tensor:
        device path, asynchronous on the current torch stream, returns tensors
        shaped ``[n_sel, 2P+1, n_obs]`` and ``[n_sel, 2P+1, 2^m]``.
        ``params``: indices of the parameters whose shifted rows are wanted (the
        others are zero or computed anyway); for sharding one circuit's shifts
        over several GPUs."""
        if stream is not None:
            with self._on_stream(stream):
                return self.psr_full(theta, inputs, sel, n_sel, shift, obs_list, n_marg_bits, None, out_expect,
                                     out_marginal, params)
        n_rows = n_sel if n_sel is not None else len(sel) if sel is not None else (1 if inputs is None else max(int(inputs.shape[0]), 1))
        plan = self._basis_plan(obs_list, n_rows * (2 * self.n_theta + 1))
        if plan is not None:
            return self._run_split(plan, "psr_full", (theta, inputs),
                                   dict(sel=sel, n_sel=n_sel, shift=shift, out_expect=out_expect,
                                        out_marginal=out_marginal, params=params), n_marg_bits)
        torch = _torch()
        L = _lib.lib()
        n_obs, M, K = len(obs_list), 1 << n_marg_bits, 2 * self.n_theta + 1
        oarr = self._obs_array(obs_list)
        B = 0 if inputs is None else int(inputs.shape[0])
        if n_sel is None:
            n_sel = len(sel) if sel is not None else max(B, 1)
        if params is not None:
            host = not isinstance(inputs, torch.Tensor) and not isinstance(theta, torch.Tensor)
            mask = np.zeros(max(self.n_theta, 1), np.uint8)
            mask[np.asarray(params, dtype=np.int64)] = 1
            dev = f"cuda:{self.device}"
            th = self.to_device(theta, np.float64)
            ia = None if inputs is None else self.to_device(inputs, np.float64)
            se = None if sel is None else self.to_device(sel, np.int32)
            oe = out_expect if out_expect is not None else torch.zeros((n_sel, K, n_obs), dtype=torch.float64, device=dev)
            om = out_marginal
            if om is None and n_marg_bits > 0:
                om = torch.zeros((n_sel, K, M), dtype=torch.float64, device=dev)
            st = torch.cuda.current_stream(self.device).cuda_stream
            p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
            _lib.check(L.vqa_psr_params_dev(self.handle(), self.device, C.c_void_p(st), p(th), p(ia), B, p(se), n_sel,
                                            float(shift), _lib.nptr(mask), oarr, n_obs, n_marg_bits,
                                            p(oe) if n_obs else None, p(om)))
            if host:
                return oe.cpu().numpy(), None if om is None else om.cpu().numpy()
            return oe, om
        if not isinstance(inputs, torch.Tensor) and not isinstance(theta, torch.Tensor):
            theta = np.ascontiguousarray(theta, np.float64)
            ia = None if inputs is None else np.ascontiguousarray(inputs, np.float64)
            se = None if sel is None else np.ascontiguousarray(sel, np.int32)
            oe = np.zeros((n_sel, K, n_obs), np.float64)
            om = np.zeros((n_sel, K, M), np.float64) if n_marg_bits > 0 else None
            _lib.check(L.vqa_psr_full(self.handle(), self.device, _lib.nptr(theta), _lib.nptr(ia), B,
                                      _lib.nptr(se), n_sel, float(shift), oarr, n_obs, n_marg_bits,
                                      _lib.nptr(oe) if n_obs else None, _lib.nptr(om)))
            return oe, om
        dev = f"cuda:{self.device}"
        th = self.to_device(theta, np.float64)
        ia = None if inputs is None else self.to_device(inputs, np.float64)
        se = None if sel is None else self.to_device(sel, np.int32)
        oe = out_expect if out_expect is not None else torch.empty((n_sel, K, n_obs), dtype=torch.float64, device=dev)
        om = out_marginal
        if om is None and n_marg_bits > 0:
            om = torch.empty((n_sel, K, M), dtype=torch.float64, device=dev)
        st = torch.cuda.current_stream(self.device).cuda_stream
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        _lib.check(L.vqa_psr_full_dev(self.handle(), self.device, C.c_void_p(st), p(th), p(ia), B,
                                      p(se), n_sel, float(shift), oarr, n_obs, n_marg_bits,
                                      p(oe) if n_obs else None, p(om)))
        return oe, om
