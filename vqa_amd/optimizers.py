"""Mini-batch optimizers that drive a QNN target object.

The reference's optimizers (vqa/VQA_optimizers/base.py, qml_optimizers.py) are
the CALLERS of the hot path, not part of it; they need only numpy and work
unchanged on the target objects of this package.  This module is the compact
counterpart shipped with the package so that ``compile`` / ``fit`` run where the
reference is not installed: the same target-object protocol
(``methods_for_optimization``, shared one-element counters; base.py:390-415),
the same step rules and bookkeeping for Adam, SGD and SGDM.  ``LR_rule`` /
``s_list_rule`` / ``batch_size_rule`` are callables (or plain constants); as in
the reference (base.py:595-615) a rule is called with the shots spent so far
unless ``<name>_annealing_by_shots`` is False, in which case it is called with
the iteration number.  The reference's power-law rule generators, suffix /
model averaging and result files are analysis tooling and are not restated.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, Optional

import numpy as np


def map_to_pi_range(x, mask_inds=None):
    """vqa/VQA_optimizers/base.py:17-24: wrap the selected entries into (-pi, pi]."""
    if mask_inds is None:
        mask_inds = []
    y = np.array(x)
    temp_y = y[mask_inds] % (2 * np.pi)
    temp_y -= 2 * np.pi * (temp_y > np.pi)
    y[mask_inds] = temp_y
    return y


def _as_rule(v) -> Callable:
    return v if callable(v) else (lambda t, _v=v: _v)


class MiniBatchOptimizer:
    """Loop and bookkeeping of vqa/VQA_optimizers/base.py:1147-1223 for the
    iteration-indexed case: counters reset before every step, periodic
    parameters wrapped after it, ``para_evol`` / ``func_evol`` / totals
    appended."""

    optimizer_name = "base"
    default_hpara: Dict = {}

    def __init__(self, hpara: Optional[dict] = None, init_spec_dict: Optional[dict] = None):
        self.set_hyper_parameters(hpara or {})
        self.initialized = False
        if init_spec_dict:
            self.initialize_optimization(init_spec_dict)

    # ---- configuration -------------------------------------------------------
    def set_hyper_parameters(self, hpara: dict) -> None:
        merged = dict(LR_rule=0.1, s_list_rule=None, batch_size_rule=16, LR_annealing_by_shots=True,
                      s_list_annealing_by_shots=True, batch_size_annealing_by_shots=True)
        merged.update(self.default_hpara)
        for k, v in merged.items():
            if not hasattr(self, k):
                setattr(self, k, v)
        for k, v in hpara.items():
            setattr(self, k, v)

    def initialize_optimization(self, init_spec_dict: dict) -> None:
        target = init_spec_dict.get("target_obj", getattr(self, "target_obj", None))
        init_params = init_spec_dict.get("init_params", getattr(self, "init_params", None))
        if target is None or init_params is None:
            return
        self.target_obj = target
        self.n_params = len(init_params)
        self.periodic_para_inds = getattr(target, "periodic_para_inds", np.arange(target.n_c_params))
        self.shots_counter = target.shots_counter
        self.call_counter = target.call_counter
        self.circuit_counter = target.circuit_counter
        self.epoch_counter = target.epoch_counter
        self.shot_time = getattr(target, "shot_time", 1.0)
        self.per_circuit_overhead = getattr(target, "per_circuit_overhead", 0.0)
        self.communication_overhead = getattr(target, "communication_overhead", 0.0)
        methods = getattr(target, "methods_for_optimization", {})
        self.grad = init_spec_dict.get("grad", methods.get("grad"))
        self.func_to_track = init_spec_dict.get("func_to_track", methods.get("func_to_track"))
        self.init_params = np.array(init_params, dtype=float)
        self.iteration_num = 0
        self.total_shots, self.total_time, self.total_call, self.total_epoch = [0], [0.0], [0], [0]
        self.shots_per_step, self.time_per_step, self.call_per_step = [0], [0.0], [0]
        self.para_evol = [self.init_params.copy()]
        self.func_evol = [self.func_to_track(self.init_params)] if self.func_to_track else [None]
        self._reset_state()
        self._last_tracked = None
        self.epoch_counter[0] = 0
        self.initialized = True

    def _reset_state(self) -> None:
        pass

    def _rule(self, name: str, t: int, floor: bool = False):
        by_shots = getattr(self, name.replace("_rule", "") + "_annealing_by_shots", True)
        v = _as_rule(getattr(self, name))(self.total_shots[-1] if by_shots else t)
        if floor:
            return np.floor(v).astype(int) if isinstance(v, np.ndarray) else int(np.floor(v))
        return v

    # ---- main loop -------------------------------------------------------------
    def new_optimize(self, init_spec_dict=None, hpara=None, options=None, **kwargs):
        if hpara:
            self.set_hyper_parameters(hpara)
        self.initialize_optimization(init_spec_dict or {})
        return self.optimize(options, **kwargs)

    def continue_optimize(self, options_to_update=None, kwargs_to_update=None):
        return self.optimize(options_to_update, **(kwargs_to_update or {}))

    def optimize(self, options: Optional[dict] = None, **kwargs):
        if not self.initialized:
            raise RuntimeError("initialize_optimization has not been given a target object and initial parameters")
        opt = dict(max_iter=None, max_shots=None, max_time=None, max_epoch=None, track_interval=1, track_loss=True,
                   track_indicator="total_shots")
        opt.update({k: v for k, v in (options or {}).items() if k in opt})
        if self._last_tracked is None:
            self._last_tracked = -opt["track_interval"]
        if all(opt[k] is None for k in ("max_iter", "max_shots", "max_time", "max_epoch")):
            opt["max_shots"] = 1e7

        def keep_going():
            spent = [self.iteration_num, self.total_shots[-1], self.total_time[-1], self.total_epoch[-1]]
            budget = [opt["max_iter"], opt["max_shots"], opt["max_time"], opt["max_epoch"]]
            return all(b is None or s < b for s, b in zip(spent, budget))

        t0 = time.time()
        new_params = self.para_evol[-1]
        while keep_going():
            self.circuit_counter[0] = 0
            self.shots_counter[0] = 0
            self.call_counter[0] = 0
            new_params = self.optimizer_step(new_params, **kwargs)
            new_params = map_to_pi_range(new_params, self.periodic_para_inds)
            self.iteration_num += 1
            self.para_evol.append(new_params)
            # base.py:1175-1196: the indicator is read before this step's totals are appended
            now = {"iteration_num": self.iteration_num, "total_shots": self.total_shots[-1],
                   "total_time": self.total_time[-1]}[opt["track_indicator"]]
            if opt["track_indicator"] == "iteration_num":
                due = self.iteration_num % opt["track_interval"] == 0
            else:
                due = now - self._last_tracked >= opt["track_interval"]
            if opt["track_loss"] and self.func_to_track and due:
                self.func_evol.append(self.func_to_track(new_params))
                self._last_tracked = now
            else:
                self.func_evol.append(None)
            shots_step = self.shots_counter[0]
            # one communication per step is counted when the target keeps no counter of its own
            time_step = (self.per_circuit_overhead * self.circuit_counter[0] + self.shot_time * shots_step
                         + self.communication_overhead * 1)
            self.shots_per_step.append(shots_step)
            self.time_per_step.append(time_step)
            self.call_per_step.append(self.call_counter[0])
            self.total_shots.append(self.total_shots[-1] + shots_step)
            self.total_time.append(self.total_time[-1] + time_step)
            self.total_call.append(self.total_call[-1] + self.call_counter[0])
            self.total_epoch.append(self.epoch_counter[0])
        if self.func_evol[-1] is None and self.func_to_track:
            self.func_evol[-1] = self.func_to_track(self.para_evol[-1])
        self.elapsed_time = time.time() - t0
        self.circuit_counter[0] = 0
        self.shots_counter[0] = 0
        func_evol = list(self.func_evol)
        for i in range(1, len(func_evol)):  # forward fill, as base.py:68-79 does for the returned lists
            if func_evol[i] is None:
                func_evol[i] = func_evol[i - 1]
        return dict(iteration_number=np.arange(self.iteration_num + 1), para_evol=self.para_evol, func_evol=func_evol,
                    total_shots=self.total_shots, total_time=self.total_time, total_call=self.total_call,
                    total_epoch=self.total_epoch, shots_per_step=self.shots_per_step, time_per_step=self.time_per_step,
                    call_per_step=self.call_per_step)

    def optimizer_step(self, params, **kwargs):
        raise NotImplementedError


class Adam(MiniBatchOptimizer):
    """vqa/VQA_optimizers/qml_optimizers.py:700-754 (beta1 .9, beta2 .99, eps 1e-8;
    rules evaluated at t+1)."""

    optimizer_name = "Adam"
    default_hpara = dict(beta1=0.9, beta2=0.99, epsilon=1e-8)

    def _reset_state(self):
        self.m_t = np.zeros(self.n_params)
        self.v_t = np.zeros(self.n_params)

    def optimizer_step(self, params, **kwargs):
        t = self.iteration_num
        self.LR_t = self._rule("LR_rule", t + 1)
        self.s_list = self._rule("s_list_rule", t + 1, floor=True)
        self.batch_size = self._rule("batch_size_rule", t + 1, floor=True)
        t += 1
        g = self.grad(params=params, n_shots_list=self.s_list, mini_batch_size=self.batch_size, **kwargs)
        self.m_t = self.beta1 * self.m_t + (1.0 - self.beta1) * g
        self.v_t = self.beta2 * self.v_t + (1.0 - self.beta2) * np.square(g)
        hat_m = self.m_t / (1.0 - self.beta1 ** t)
        hat_v = self.v_t / (1.0 - self.beta2 ** t)
        return params - self.LR_t * hat_m / (np.sqrt(hat_v) + self.epsilon)


class SGD(MiniBatchOptimizer):
    """vqa/VQA_optimizers/qml_optimizers.py:942-979 (rules evaluated at t)."""

    optimizer_name = "vanilla SGD"

    def optimizer_step(self, params, **kwargs):
        t = self.iteration_num
        self.LR_t = self._rule("LR_rule", t)
        self.s_list = self._rule("s_list_rule", t, floor=True)
        self.batch_size = self._rule("batch_size_rule", t, floor=True)
        g = self.grad(params=params, n_shots_list=self.s_list, mini_batch_size=self.batch_size, **kwargs)
        return params - self.LR_t * g


class SGDM(MiniBatchOptimizer):
    """vqa/VQA_optimizers/qml_optimizers.py:981-1027 (mu .9; LR and shots at t+1,
    batch size at t)."""

    optimizer_name = "SGDM"
    default_hpara = dict(mu=0.9)

    def _reset_state(self):
        self.m_t = np.zeros(self.n_params)

    def optimizer_step(self, params, **kwargs):
        t = self.iteration_num
        self.LR_t = self._rule("LR_rule", t + 1)
        self.s_list = self._rule("s_list_rule", t + 1, floor=True)
        self.batch_size = self._rule("batch_size_rule", t, floor=True)
        g = self.grad(params=params, n_shots_list=self.s_list, mini_batch_size=self.batch_size, **kwargs)
        self.m_t = self.mu * self.m_t + g
        return params - self.LR_t * self.m_t


_BUILTIN = {"Adam": Adam, "SGD": SGD, "SGDM": SGDM}


def get_optimizer_class(name, prefer_reference: bool = False):
    """Optimizer class by name (a class is passed through).  Built-in: Adam, SGD,
    SGDM.  Any other name, or every name with ``prefer_reference``, is looked
    up in the reference's ``vqa.VQA_optimizers.qml_optimizers`` when that
    package is importable (SantaQlaus, gCANS, SPSA, Adam_ASS live only there)."""
    if not isinstance(name, str):
        return name
    if name in _BUILTIN and not prefer_reference:
        return _BUILTIN[name]
    try:
        from vqa.VQA_optimizers import qml_optimizers as ref  # type: ignore

        if hasattr(ref, name):
            return getattr(ref, name)
    except ImportError:
        pass
    if name in _BUILTIN:
        return _BUILTIN[name]
    raise ValueError(f"optimizer {name!r} needs the reference's vqa.VQA_optimizers package on the path; "
                     f"built in: {sorted(_BUILTIN)}")
