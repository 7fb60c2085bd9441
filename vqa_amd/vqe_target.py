"""Pauli-sum (VQE) target object on the batched HIP engine.

The reference ships optimizers for variational eigensolvers
(vqa/VQA_optimizers/vqe_optimizers.py) but no target object for them; this
class supplies one that satisfies their protocol: ``grad(params, s_list)``
(vqe_optimizers.py:637, 863, 908), ``iEvaluate(params, s_list) -> {'grad',
'gvar_list'}`` (371, 701, 970-977), ``loss(params, n_shots)`` (584-585, 973),
``func_to_track(params)``, ``n_params`` / ``n_c_params``, the one-element-list
counters and the overhead times (VQA_optimizers/base.py:390-415).  Its
statistics are this package's own (there is nothing in the reference to match):

* H = sum_t c_t P_t; every Pauli term is measured on its own, the n shots of an
  evaluation are dealt to the terms in proportion to |c_t| (at least one each);
* exact mode (``min_approx_shots == -10``): every estimate is the expectation
  value; otherwise the estimate of term t is ``E_t + randn() sqrt((1 - E_t^2) /
  n_t)`` (NumPy's global generator; order: evaluation-major, term-minor);
* gradient by the parameter-shift rule, ``g_i = (E(theta + pi/2 e_i) - E(theta
  - pi/2 e_i)) / 2`` with ``s_list[i]`` shots for each of the two evaluations;
  ``gvar_list[i]`` is the variance of one shot's contribution to g_i, i.e.
  ``s_list[i] * Var(g_i)``, from the exact term variances.

All 2P+1 circuits of a gradient go to the GPU as one instance list with one
observable per term (``LearningCircuit.eval_batch``); terms with X or Y are
served by the engines that measure general Pauli strings.
"""
from __future__ import annotations

import uuid

import numpy as np

from .observable import Observable, PauliTerm
from .qnn_base import _np


class PauliSumVQE:
    def __init__(self, n_qubits, learning_circuit, hamiltonian: Observable, min_approx_shots=0, setting_dict=None,
                 id_string="VQE", file_id=None, shot_time=1.0, per_circuit_overhead=0.0, communication_overhead=0.0,
                 x=None):
        self.n_qubits = n_qubits
        self.circuit = learning_circuit
        self.n_c_params = len(self.circuit.get_parameters())
        self.n_params = self.n_c_params
        self.periodic_para_inds = np.arange(self.n_params)
        self.min_approx_shots = min_approx_shots
        self.call_counter, self.shots_counter, self.circuit_counter, self.epoch_counter = [0], [0], [0], [0]
        self.shot_time, self.per_circuit_overhead = shot_time, per_circuit_overhead
        self.communication_overhead = communication_overhead
        self.id_string = id_string
        self.file_id = file_id if file_id else str(uuid.uuid4())
        self.results_dict = {}
        # one observable per Pauli term (coefficient 1), the identity part kept aside
        self.coefs, self.offset, self.terms = [], 0.0, []
        for t in hamiltonian.terms:
            if not t.paulis:
                self.offset += t.coef
                continue
            o = Observable(n_qubits)
            o.terms.append(PauliTerm(1.0, t.paulis))
            self.terms.append(o)
            self.coefs.append(t.coef)
        self.coefs = np.array(self.coefs)
        self.hamiltonian = hamiltonian
        # circuits without input gates take one dummy sample
        self._x = np.zeros((1, 1)) if x is None else np.atleast_2d(np.asarray(x, dtype=float))
        self._inputs = None
        self.group = None   # a torch.distributed process group: shard the shifted evaluations over its ranks
        self.methods_for_optimization = {}
        self.set_methods_for_optimizer(setting_dict or {
            "func_to_track": "exact_energy", "loss": "energy_eval", "grad": "grad_eval",
            "iEvaluate": "iEval", "iEvaluate_iRandom": "iEval"})

    def set_methods_for_optimizer(self, setting_dict):
        for key, name in setting_dict.items():
            self.methods_for_optimization[key] = getattr(self, name)

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_inputs"] = None
        d["terms"] = None
        d["group"] = None
        d["methods_for_optimization"] = {k: v.__name__ for k, v in self.methods_for_optimization.items()}
        return d

    # ---- circuit evaluation -----------------------------------------------------------
    def _term_expectations(self, params, shift_param, shift_val):
        """Exact <P_t> of every term for each instance: [E, n_terms]."""
        if self._inputs is None:
            self._inputs = self.circuit.upload_inputs(self._x)
        theta = np.asarray(params[:self.n_c_params], dtype=np.float64)
        E = len(shift_param)
        si = np.zeros(E, np.int32)
        out = np.empty((E, len(self.terms)))
        for lo in range(0, len(self.terms), 64):      # the library takes up to 64 observables per call
            e, _ = self.circuit.eval_batch(theta, self._inputs, si, np.asarray(shift_param, np.int32),
                                           np.asarray(shift_val, np.float64), obs_list=self.terms[lo:lo + 64])
            out[:, lo:lo + 64] = _np(e)
        self.circuit_counter[0] += E
        self.call_counter[0] += E
        return out

    def _deal_shots(self, n):
        """Shots of one evaluation per term, in proportion to |c_t|, at least one each."""
        w = np.abs(self.coefs) / np.sum(np.abs(self.coefs))
        return np.maximum(1, np.floor(np.asarray(n, dtype=float)[..., None] * w)).astype(np.int64)

    def _estimate_energy(self, Et, n):
        """Energy estimates of the instances with exact term values Et [E, T] and n [E] shots each."""
        if self.min_approx_shots == -10:
            return self.offset + Et @ self.coefs
        nt = self._deal_shots(n)
        self.shots_counter[0] += int(nt.sum())
        est = Et + self._noise(Et.shape) * np.sqrt(np.clip(1.0 - Et ** 2, 0.0, None) / nt)
        return self.offset + est @ self.coefs

    def _noise(self, shape):
        """Standard normal deviates from NumPy's global generator.  With the parameters sharded over
        a process group every rank must see the SAME estimates (or the ranks' parameters drift
        apart): rank 0 draws and broadcasts; the other ranks draw as well and discard, so that
        their generators advance alike."""
        z = np.random.randn(*shape)
        group = getattr(self, "group", None)
        if group is None:
            return z
        import torch
        import torch.distributed as dist

        if dist.get_world_size(group) == 1:
            return z
        dev = getattr(self._inputs, "device", "cpu") if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.from_numpy(z).to(dev)
        dist.broadcast(t, src=dist.get_global_rank(group, 0), group=group)
        return t.cpu().numpy()

    # ---- the optimizer-facing methods ----------------------------------------------------
    def exact_energy(self, params):
        Et = self._term_expectations(params, [-1], [0.0])
        return float(self.offset + Et[0] @ self.coefs)

    def energy_eval(self, params, n_shots, **unused):
        Et = self._term_expectations(params, [-1], [0.0])
        return float(self._estimate_energy(Et, np.array([n_shots]))[0])

    def _shift_lists(self):
        P = self.n_c_params
        sp = np.repeat(np.arange(P, dtype=np.int32), 2)
        sv = np.tile([0.5 * np.pi, -0.5 * np.pi], P)
        return sp, sv

    def grad_eval(self, params, n_shots_list, **unused):
        return self.iEval(params, n_shots_list)["grad"]

    def _shifted_term_expectations(self, params):
        """Exact <P_t> at theta_i +- pi/2 for every parameter, [2P, T] (+ first, - second): one full
        parameter-shift sweep on the device (the engines share work between the shifts of a
        sweep; an explicit instance list would start every shifted circuit from |0>).

        With ``self.group`` set to a torch.distributed process group (one process per GPU) the
        parameters are dealt round-robin to the ranks: each rank sweeps its own parameters and
        one all-reduce (sum of matrices that are zero outside a rank's rows) puts the whole
        [2P, T] matrix on every rank."""
        if not hasattr(self.circuit, "psr_full"):
            sp, sv = self._shift_lists()
            return self._term_expectations(params, sp, sv)
        if self._inputs is None:
            self._inputs = self.circuit.upload_inputs(self._x)
        theta = np.asarray(params[:self.n_c_params], dtype=np.float64)
        P = self.n_c_params
        rank, world = 0, 1
        group = getattr(self, "group", None)
        if group is not None:
            import torch.distributed as dist

            rank, world = dist.get_rank(group), dist.get_world_size(group)
        mine = np.arange(rank, P, world)
        out = np.zeros((2 * P, len(self.terms)))
        rows = np.concatenate([2 * mine, 2 * mine + 1])
        for lo in range(0, len(self.terms), 64):      # the library takes up to 64 observables per call
            e, _ = self.circuit.psr_full(theta, self._inputs, n_sel=1, obs_list=self.terms[lo:lo + 64],
                                         params=None if world == 1 else mine)
            out[rows, lo:lo + 64] = _np(e)[0, 1:, :][rows]
        if world > 1:
            import torch

            from .distributed import all_reduce_sum

            dev = getattr(self._inputs, "device", "cpu")
            t = torch.from_numpy(out).to(dev)
            all_reduce_sum(t, group=group)
            out = t.cpu().numpy()
        # counters: every rank accounts for the whole sweep (they describe the job, not the rank)
        self.circuit_counter[0] += 2 * P
        self.call_counter[0] += 2 * P
        return out

    def iEval(self, params, n_shots_list, **unused):
        P = self.n_c_params
        Et = self._shifted_term_expectations(params)                    # [2P, T]
        n = np.repeat(np.asarray(n_shots_list, dtype=float)[:P], 2)
        e = self._estimate_energy(Et, n)
        grad = 0.5 * (e[0::2] - e[1::2])
        # variance of one shot's contribution: n_i * Var(g_i) with n_t = n_i |c_t| / sum|c|
        var_terms = np.clip(1.0 - Et ** 2, 0.0, None) @ np.abs(self.coefs) * np.sum(np.abs(self.coefs))
        gvar = 0.25 * (var_terms[0::2] + var_terms[1::2])
        return {"grad": grad, "gvar_list": gvar}
