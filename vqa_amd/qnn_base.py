"""QNN regressor target object on the batched HIP engine.

Drop-in counterpart of ``myQNN_base`` (vqa/QNN/myQNNs_base.py:264-1675): same
constructor, same optimizer-facing protocol (``methods_for_optimization``,
``n_params`` / ``n_c_params``, one-element-list counters, overhead times) and
the same estimator algebra.  What changes is how circuits are evaluated: where
the reference loops ``circuit.run(x)`` over samples, shifts and parameters
(myQNNs_base.py:1110-1124, 1381-1394), this class asks the circuit object for
one batched sweep (``psr_full`` / ``eval_batch`` of vqa_amd.LearningCircuit)
and applies the reference's per-sample statistics to the returned matrix of
exact expectation values with NumPy:

* exact mode (``min_approx_shots == -10``), CLT mode (``n >= min_approx_shots``:
  ``E + randn() * sqrt((1 - E^2) / n)``) and finite-shot mode (the reference draws
  basis states with qulacs' private generator, so this mode agrees with it in
  distribution only).  Finite-shot draws come in two flavours, ``shot_draws``:
  ``"binomial"`` (default; one binomial per sample on the exact Z0 distribution) and
  ``"per_shot"`` (every shot drawn by itself, per sample in the reference's call order, as
  ``numpy.random.choice(4, size=n, p=<distribution of qubits 0, 1>)``: the global generator
  is then consumed exactly as by the reference's per-sample code running on a
  ``state.sampling`` that draws that way, which is how tests/golden/host_layer.json -- recorded
  from the reference's own classes -- is replayed number by number);
* the NumPy global generator is consumed in the reference's order: unshifted
  evaluation first (sample-major), then per parameter the +shift samples and
  the -shift samples;
* ``shots_counter`` / ``call_counter`` / ``circuit_counter`` / ``epoch_counter``
  advance by the same totals as the per-sample code.
"""
from __future__ import annotations

import copy
import os
import pickle

try:  # the reference writes and reads its files with dill (myQNNs_base.py:443, 499): same format when it is there
    import dill as _serializer
except ImportError:  # plain pickle streams are a subset of what dill reads
    _serializer = pickle
import uuid
from typing import Optional

import numpy as np

from .observable import Observable, z_observable
from . import optimizers as _optimizers


def _np(a):
    """numpy view of a result that may be a torch tensor on the GPU."""
    if a is None:
        return None
    if hasattr(a, "detach"):
        return a.detach().cpu().numpy()
    return np.asarray(a)


def split_array_equally(arr):
    """vqa/QNN/myQNNs_base.py:234-245."""
    mid = len(arr) // 2
    return arr[:mid], arr[mid:2 * mid]


class myQNN_base:
    """Regressor: prediction = params[-1] * <Z_0>; the last parameter is a
    trainable scale.  See the module docstring for what is and is not the
    reference's code path."""

    def __init__(self, n_qubits, learning_circuit, x_train=None, y_train=None, x_test=None, y_test=None,
                 min_approx_shots=-1, setting_dict=None, id_string="QNN", file_id=None, obs=None,
                 shot_time=1.0, per_circuit_overhead=0.0, communication_overhead=0.0,
                 max_sub_nqubits_sampling_aft_partial_tr=6, device_statistics=False, shot_draws="binomial"):
        if shot_draws not in ("binomial", "per_shot"):
            raise ValueError("shot_draws must be 'binomial' or 'per_shot'")
        self.shot_draws = shot_draws
        self.circuit = learning_circuit
        # True: in pure CLT mode the gradient oracle draws its normal deviates on the GPU
        # (torch generator seeded from NumPy's) and returns only the gradient; same
        # statistics and counters, but not the reference's NumPy draw sequence
        self.device_statistics = device_statistics
        # a torch.distributed process group (one process per GPU): the mini-batch of the gradient oracles
        # that reduce on the device (exact mode, or device_statistics) is then split into contiguous blocks
        # over the group's ranks and the batch sums are added with one all-reduce; every rank holds the
        # whole data set and returns the same result.  None: one process does everything.
        self.group = None
        self.n_c_params = len(self.circuit.get_parameters())
        self.n_params = self.n_c_params + 1  # + the re-normalisation coefficient
        self.n_qubits = n_qubits
        self._handles = {}
        self.set_data(x_train, y_train, x_test, y_test)
        self.methods_for_optimization = {}
        if setting_dict:
            self.set_methods_for_optimizer(setting_dict)
        self.min_approx_shots = min_approx_shots
        self.call_counter = [0]
        self.shots_counter = [0]
        self.circuit_counter = [0]
        self.epoch_counter = [0]
        self.communication_overhead = communication_overhead
        self.shot_time = shot_time
        self.per_circuit_overhead = per_circuit_overhead
        self.id_string = id_string
        self.results_dict = {}
        self.file_id = file_id if file_id else str(uuid.uuid4())
        self.exact_cost_evol = []
        self.remain_ind = []
        self._dim = 2 ** self.n_qubits
        if obs is None:
            obs = self._default_obs()
        # as in the reference, a user observable is only serialised; the model always reads Z_0
        self.obs_json = obs.to_json()
        self.target_indices_list = [obs.get_term(i).get_index_list() for i in range(obs.get_term_count())]
        self.max_sub_nqubits_sampling_aft_partial_tr = max_sub_nqubits_sampling_aft_partial_tr
        self.n_shots_test = 1e2

    # ---- bookkeeping identical to the reference ---------------------------------
    def get_parameter_count(self, return_circuit_parameter_count=False):
        return self.n_c_params if return_circuit_parameter_count else self.n_params

    def set_data(self, x_train, y_train, x_test, y_test):
        if x_train is not None and y_train is not None:
            self.x_train = np.array(x_train)
            self.y_train = np.array(y_train)
            self.num_train_data = len(x_train)
            self._uniform_wt = np.full(len(self.x_train), 1.0)
            self._uniform_wt_dist = np.full(len(self.x_train), 1.0) / len(self.x_train)
            self._handles.pop("train", None)
        if x_test is not None and y_test is not None:
            self.x_test = np.array(x_test)
            self.y_test = np.array(y_test)
            self._handles.pop("test", None)

    def _default_obs(self):
        return z_observable(self.n_qubits, 0)

    def remove_data(self):
        for a in ("x_train", "y_train", "num_train_data", "x_test", "y_test"):
            if hasattr(self, a):
                delattr(self, a)
        self._handles = {}

    def get_data(self):
        return (getattr(self, "x_train", []), getattr(self, "y_train", []),
                getattr(self, "x_test", []), getattr(self, "y_test", []))

    def set_overhead_time(self, shot_time=None, per_circuit_overhead=None, communication_overhead=None):
        if shot_time is not None:
            self.shot_time = shot_time
        if per_circuit_overhead is not None:
            self.per_circuit_overhead = per_circuit_overhead
        if communication_overhead is not None:
            self.communication_overhead = communication_overhead

    def set_methods_for_optimizer(self, setting_dict):
        for key, method_name in setting_dict.items():
            self.methods_for_optimization[key] = getattr(self, method_name)

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_handles"] = {}  # device-resident input matrices are rebuilt on demand
        d["_z0"] = None     # holds a native handle
        d["group"] = None   # process groups do not travel
        d["methods_for_optimization"] = {k: v.__name__ for k, v in self.methods_for_optimization.items()}
        return d

    def __setstate__(self, d):
        names = d.pop("methods_for_optimization", {})
        self.__dict__.update(d)
        self.methods_for_optimization = {k: getattr(self, v) for k, v in names.items()}

    # ---- compile / fit (myQNNs_base.py:540-690) ------------------------------------
    _LOSS_METHODS = {
        "mean_squared_error": {
            "func_to_track": "MSE_both_exact_loss_to_track", "loss": "MSE_loss_eval", "grad": "MSE_grad_eval",
            "iEvaluate": "MSE_iEval", "iEvaluate_iRandom": "MSE_iEval_iRandom",
        },
    }

    def compile(self, optimizer="SGD", loss="mean_squared_error", hpara=None, init_params=None, min_approx_shots=None,
                x_train=None, y_train=None, x_test=None, y_test=None, shot_time=None, per_circuit_overhead=None,
                communication_overhead=None, init_spec_dict=None, manual_methods_setting_dict=None):
        self.set_data(x_train, y_train, x_test, y_test)
        self._optimizer_name = optimizer if isinstance(optimizer, str) else optimizer.__name__
        if min_approx_shots is not None:
            self.min_approx_shots = min_approx_shots
        optimizer_class = _optimizers.get_optimizer_class(optimizer)
        if init_spec_dict is None:
            init_spec_dict = {}
        if init_params is not None:
            init_spec_dict["init_params"] = init_params
            self.model_params = init_params
        self.set_overhead_time(shot_time, per_circuit_overhead, communication_overhead)
        if manual_methods_setting_dict is not None:
            self.set_methods_for_optimizer(manual_methods_setting_dict)
        elif loss in self._LOSS_METHODS:
            self.set_methods_for_optimizer(self._LOSS_METHODS[loss])
        else:
            raise ValueError(f"loss {loss!r} is not available for {type(self).__name__}")
        init_spec_dict["target_obj"] = self
        self.optimizer = optimizer_class(hpara=hpara, init_spec_dict=init_spec_dict)
        self._optimized = False

    def fit(self, init_params=None, options=None, hpara=None, init_spec_dict=None, Return_results=True,
            store_results_in_self=False, auto_save=True, save_self_obj=True, save_directory="single_optimization",
            process_id=0, **kwargs):
        if not hasattr(self, "optimizer"):
            print("Optimizer has not been specified")
            return
        if options is None:
            options = {}
        options["process_id"] = process_id
        if not self._optimized:
            if init_spec_dict is None:
                init_spec_dict = {}
            if init_params is not None:
                init_spec_dict["init_params"] = init_params
                self.model_params = init_params
            init_spec_dict["target_obj"] = self
            results = self.optimizer.new_optimize(init_spec_dict=init_spec_dict, hpara=hpara, options=options, **kwargs)
        else:
            results = self.optimizer.continue_optimize(options_to_update=options, kwargs_to_update=kwargs)
        self.model_params = results["para_evol"][-1]
        if store_results_in_self:
            self.results_dict.setdefault(self._optimizer_name, []).append(results)
        self.elapsed_time_for_optimization = getattr(self.optimizer, "elapsed_time", None)
        self._optimized = True
        if auto_save:
            self.save_to_file(results=results, save_self_obj=save_self_obj, directory=save_directory)
        return results

    def save_to_file(self, results=None, save_self_obj=True, directory=None):
        """myQNNs_base.py:404-451.  Written with dill when it is importable (the reference's
        load_from_file reads such a file), else with pickle; device handles are dropped by
        __getstate__ and come back on first use."""
        name = f"{self.id_string}_model_{getattr(self, '_optimizer_name', 'none')}_{self.file_id}.pkl"
        if directory:
            os.makedirs(directory, exist_ok=True)
            name = os.path.join(directory, name)
        data = {"results": results}
        if save_self_obj:
            try:
                shadow = copy.copy(self)
                shadow.__dict__.pop("optimizer", None)
                if _serializer is pickle:
                    shadow.circuit = None  # input_func lambdas do not pickle; the circuit is rebuilt by its builder
                _serializer.dumps(shadow)
                data["object"] = shadow
            except Exception as e:  # same policy as the reference: fall back to the results
                print(f"An error occurred while saving self object: {e}")
        with open(name, "wb") as f:
            _serializer.dump(data, f)
        return name

    def load_from_file(self, filename, directory="."):
        """myQNNs_base.py:473-507: a saved object replaces this instance's state (the circuit stays
        when the file holds none), saved results alone go to latest_results."""
        try:
            with open(os.path.join(directory, filename), "rb") as f:
                loaded = _serializer.load(f)
            if "object" in loaded:
                state = dict(loaded["object"].__dict__)
                if state.get("circuit") is None:
                    state.pop("circuit", None)
                self.__dict__.update(state)
                print(f"Self object updated from the file: {filename}")
            if loaded.get("results") is not None:
                self.latest_results = loaded["results"]
                if "object" not in loaded:
                    print(f"Results loaded into 'latest_results' from the file: {filename}")
        except Exception as e:
            print(f"An error occurred while loading data from file: {e}")

    # ---- batched circuit evaluation ------------------------------------------------
    def _obs_list(self):
        if getattr(self, "_z0", None) is None:
            self._z0 = self._default_obs()
        return [self._z0]

    def _handle(self, which):
        """Input-angle matrix of the train / test set, resident where the circuit wants it."""
        if which not in self._handles:
            x = self.x_train if which == "train" else self.x_test
            self._handles[which] = self.circuit.upload_inputs(x)
        return self._handles[which]

    # finite-shot draws of this class read the readout distribution, not the model observable
    # (they coincide for the regressor: both are Z0)
    _sample_from_marginal = False

    def _marg_bits(self):
        return min(2, self.n_qubits)

    @staticmethod
    def _pad4(m):
        """Distribution of qubits (0, 1) as 4 columns (a 1-qubit register has 2)."""
        m = _np(m)
        if m.shape[-1] == 4:
            return m
        out = np.zeros(m.shape[:-1] + (4,))
        out[..., :m.shape[-1]] = m
        return out

    def _may_sample(self, n_s, min_approx):
        """Whether any estimate of a call is drawn shot by shot (myQNNs_base.py:806, 883)."""
        if min_approx == -10:
            return False
        n_s = np.asarray(n_s)
        return bool(min_approx < 0 or (n_s.size and n_s.min() < min_approx))

    def _wants_marginal(self, n_s, min_approx, no_power=False):
        sampling = self._may_sample(n_s, min_approx) or (no_power and min_approx == -10)
        return sampling and (self.shot_draws == "per_shot" or self._sample_from_marginal)

    def _expect_all(self, params, x_list=None, which=None, idx=None, with_marginal=False):
        """Exact <Z0> for each sample: an explicit x_list, or rows idx of a resident set.
        with_marginal: also the distribution of qubits (0, 1), [B, 4]."""
        theta = np.asarray(params[:self.n_c_params], dtype=np.float64)
        mb = self._marg_bits() if with_marginal else 0
        if x_list is not None:
            inputs = self.circuit.upload_inputs(np.asarray(x_list))
            e, m = self.circuit.eval_batch(theta, inputs, obs_list=self._obs_list(), n_marg_bits=mb)
        else:
            e, m = self.circuit.eval_batch(theta, self._handle(which), sample_idx=None if idx is None else np.asarray(idx, np.int32),
                                           obs_list=self._obs_list(), n_marg_bits=mb)
        if with_marginal:
            return _np(e)[:, 0], self._pad4(m)
        return _np(e)[:, 0]

    def _psr_all(self, params, idx, keep_on_device=False, with_marginal=False):
        """Exact <Z0> of the 2P+1 parameter-shift instances of each selected training sample: [B, 2P+1].
        with_marginal: also the distribution of qubits (0, 1) of every instance, [B, 2P+1, 4]."""
        theta = np.asarray(params[:self.n_c_params], dtype=np.float64)
        mb = self._marg_bits() if with_marginal else 0
        e, m = self.circuit.psr_full(theta, self._handle("train"), sel=np.asarray(idx, np.int32), obs_list=self._obs_list(),
                                     n_marg_bits=mb)
        if with_marginal:
            return _np(e)[:, :, 0], self._pad4(m)
        return e[:, :, 0] if keep_on_device else _np(e)[:, :, 0]

    # ---- per-sample statistics (myQNNs_base.py:715-738, 799-905) ----------------------
    def _unbiased_powerE_Pauli(self, sample_mean, n_samples, order=4, unbiased=False):
        """myQNNs_base.py:820-865, elementwise over samples."""
        n = np.asarray(n_samples, dtype=float)
        M = np.asarray(sample_mean, dtype=float)
        out = [M]
        if order >= 2:
            out.append((n / (n - 1.0)) * M ** 2 - 1 / (n - 1.0) if unbiased else M ** 2)
        if order >= 3:
            out.append((n / (n - 1.0)) * (n / (n - 2.0)) * M ** 3 - (3.0 * n - 2) / (n - 1.0) / (n - 2.0) * M
                       if unbiased else M ** 3)
        if order >= 4:
            out.append((n / (n - 1.0)) * (n / (n - 2.0)) * (n / (n - 3.0)) * M ** 4
                       - ((n / (n - 1)) * ((6.0 * n - 8.0) / (n - 2.0)) * M ** 2 - 3.0 / (n - 1.0)) / (n - 3.0)
                       if unbiased else M ** 4)
        return out[:max(order, 1)]

    @staticmethod
    def _shots_low2(marg_row, n):
        """n shots of qubits (0, 1), one uniform each, from the global generator."""
        return np.random.choice(4, size=int(n), p=marg_row)

    def _draw_means(self, exact, n_s, min_approx, marg=None):
        """Estimated <Z0> of each sample from its exact value: CLT noise or finite-shot
        sampling per sample, global NumPy generator, sample order.  marg [B, 4]: the
        distribution of qubits (0, 1), which the finite-shot draws read when given."""
        exact = np.asarray(exact, dtype=float)
        n_s = np.asarray(n_s)
        B = len(exact)
        sampled = (n_s < min_approx) | (min_approx < 0)
        self.call_counter[0] += B + int(sampled.sum())
        self.shots_counter[0] += int(n_s.sum())
        var = 1.0 - exact ** 2
        with np.errstate(divide="ignore", invalid="ignore"):
            if not sampled.any():
                return exact + np.random.randn(B) * np.sqrt(var / n_s)
            est = np.empty(B)
            if self.shot_draws == "per_shot":
                if marg is None:
                    raise ValueError("per-shot draws need the readout distribution")
                for j in range(B):
                    if sampled[j]:
                        est[j] = np.mean((-1) ** (self._shots_low2(marg[j], n_s[j]) % 2))    # myQNNs_base.py:725-733
                    else:
                        est[j] = exact[j] + np.random.randn() * np.sqrt(var[j] / n_s[j])
                return est
            p_minus = np.clip(0.5 * (1.0 - exact) if marg is None else marg[:, 1] + marg[:, 3], 0.0, 1.0)
            if sampled.all():
                k = np.random.binomial(n_s.astype(np.int64), p_minus)
                return (n_s - 2.0 * k) / n_s
            for j in range(B):  # mixed modes: keep the generator in sample order
                if sampled[j]:
                    est[j] = (n_s[j] - 2.0 * np.random.binomial(int(n_s[j]), p_minus[j])) / n_s[j]
                else:
                    est[j] = exact[j] + np.random.randn() * np.sqrt(var[j] / n_s[j])
            return est

    def _estimate(self, exact, n_s, min_approx, order=4, Return_var=False, unbiased=False, no_power=False, marg=None):
        """Batched _single_state_hybrid_sampler[_power] (myQNNs_base.py:799-818, 867-905)."""
        exact = np.asarray(exact, dtype=float)
        n_s = np.asarray(n_s)
        self.circuit_counter[0] += len(exact)
        if min_approx == -10 and not no_power:
            self.call_counter[0] += len(exact)
            EO = [exact ** k for k in range(1, order + 1)]
            return (EO, 1.0 - exact ** 2) if Return_var else EO
        M = self._draw_means(exact, n_s, min_approx, marg)
        nf = n_s.astype(float)
        with np.errstate(divide="ignore", invalid="ignore"):
            var = nf / (nf - 1.0) * (1.0 - M ** 2)
            if no_power:
                return (M, var) if Return_var else M
            EO = self._unbiased_powerE_Pauli(M, nf, order=order, unbiased=unbiased)
        return (EO, var) if Return_var else EO

    # the reference's per-call names, batched ------------------------------------------
    def _sample_obs_data_by_data(self, params, x_eval, n_s_list, min_approx, order=4, Return_var=False, unbiased=False):
        """myQNNs_base.py:1100-1125 for an explicit list of samples."""
        exact, marg = self._expect_maybe_marg(params, n_s_list, min_approx, x_list=x_eval)
        return self._estimate(exact, n_s_list, min_approx, order=order, Return_var=Return_var, unbiased=unbiased, marg=marg)

    def _expect_maybe_marg(self, params, n_s, min_approx, no_power=False, **where):
        """(exact, marg or None): the readout distribution rides along when draws need it."""
        if self._wants_marginal(n_s, min_approx, no_power):
            return self._expect_all(params, with_marginal=True, **where)
        return self._expect_all(params, **where), None

    def _sample_predict_scaled(self, x_eval, n_s_list, params=None, min_approx=None, Return_var=False):
        """myQNNs_base.py:1044-1070."""
        if min_approx is None:
            min_approx = self.min_approx_shots
        if params is None:
            params = self.model_params
        exact, marg = self._expect_maybe_marg(params, n_s_list, min_approx, no_power=True, x_list=x_eval)
        w = params[-1]
        if Return_var:
            M, var = self._estimate(exact, n_s_list, min_approx, Return_var=True, no_power=True, marg=marg)
            return w * M, var * w ** 2
        return self._estimate(exact, n_s_list, min_approx, no_power=True, marg=marg) * w

    def _sample_predict_no_scale(self, x_eval, n_s_list, params=None, min_approx=None, Return_var=False):
        """myQNNs_base.py:1072-1097."""
        if min_approx is None:
            min_approx = self.min_approx_shots
        if params is None:
            params = self.model_params
        exact, marg = self._expect_maybe_marg(params, n_s_list, min_approx, no_power=True, x_list=x_eval)
        return self._estimate(exact, n_s_list, min_approx, Return_var=Return_var, no_power=True, marg=marg)

    def sample_predict(self, x_eval, n_s_list, params=None, min_approx=None, Return_var=False):
        return self._sample_predict_scaled(x_eval, n_s_list, params, min_approx, Return_var)

    # ---- exact forward (myQNNs_base.py:907-994) -----------------------------------------
    def exact_predict(self, x_list, params=None):
        if params is None:
            params = self.model_params
        return params[-1] * self._expect_all(params, x_list=x_list)

    def _exact_predict_set(self, which, params):
        return params[-1] * self._expect_all(params, which=which)

    def exact_MSE(self, params, which="test", ret_both=False, log_evol=False):
        if ret_both:
            which = "test"
        if which == "train":
            y_pred = self._exact_predict_set("train", params).astype(float)
            MSE = np.sum(np.square(y_pred - self.y_train)) / len(self.y_train)
            if log_evol:
                self.exact_cost_evol.append(self.exact_MSE(params, which="test"))
            return MSE
        y_pred = self._exact_predict_set("test", params).astype(float)
        MSE = np.sum(np.square(y_pred - self.y_test)) / len(self.y_test)
        if log_evol:
            self.exact_cost_evol.append(MSE)
        if ret_both:
            return {"test": MSE, "train": self.exact_MSE(params, which="train", ret_both=False)}
        return MSE

    def exact_BMA_predict(self, x_list, params_list, num_avg):
        num_avg = min(len(params_list), num_avg)
        return np.mean(np.array([self.exact_predict(x_list, p) for p in params_list[-num_avg:]]), axis=0)

    def exact_BMA_MSE(self, params_list, num_avg, which="test", ret_both=False):
        if ret_both:
            which = "test"
        if which == "train":
            y_pred = self.exact_BMA_predict(self.x_train, params_list, num_avg).astype(float)
            return np.sum(np.square(y_pred - self.y_train)) / len(self.y_train)
        y_pred = self.exact_BMA_predict(self.x_test, params_list, num_avg).astype(float)
        MSE = np.sum(np.square(y_pred - self.y_test)) / len(self.y_test)
        if ret_both:
            return {"test": MSE, "train": self.exact_BMA_MSE(params_list, num_avg, which="train")}
        return MSE

    def MSE_exact_train_loss(self, params):
        return self.exact_MSE(params, which="train", log_evol=True)

    def MSE_both_exact_loss_to_track(self, params):
        return self.exact_MSE(params, which="test", ret_both=True, log_evol=False)

    def reset_exact_cost_evol(self):
        self.exact_cost_evol = []

    def generate_labels(self, x_data, target_params):
        return self.exact_predict(x_data, target_params)

    # ---- variance estimators (myQNNs_base.py:1205-1316, 1599-1617) -----------------------
    def _estimate_MSEgrad_mean_condVar(self, y_eval, E0pow_data_list, w=1.0, second_order=True, n0=None,
                                       Scale_para=True, Eppow_data_list=None, Empow_data_list=None,
                                       var_0=None, var_p=None, var_m=None):
        EX = np.array(E0pow_data_list[0])
        EX_2 = np.array(E0pow_data_list[1])
        if Scale_para:
            EX_3 = np.array(E0pow_data_list[2])
            EX_4 = np.array(E0pow_data_list[3])
            a = -2 * y_eval
            b = 2 * w
            est = a ** 2 + 4.0 * a * b * EX + (4.0 * b ** 2 - a ** 2) * EX_2 - 4 * a * b * EX_3 - 4 * b ** 2 * EX_4
            if second_order and n0:
                est += 2 * b ** 2 * (1 - 2 * EX + EX_2) / (n0 - 1)
            mean_condVar = np.mean(est)
            if mean_condVar < 0:
                Var_X = var_0
                est = np.square(a + 2 * b * EX) * Var_X
                if second_order and n0:
                    est += 2 * b ** 2 * np.square(Var_X) / (n0 - 1)
                mean_condVar = np.mean(est)
            return mean_condVar
        EXp = np.array(Eppow_data_list[0])
        EXm = np.array(Empow_data_list[0])
        EXp_2 = np.array(Eppow_data_list[1])
        EXm_2 = np.array(Empow_data_list[1])
        w2 = w ** 2
        y2 = np.square(y_eval)
        yw = w * y_eval
        err2 = y2 - 2.0 * yw * EX + w2 * EX_2
        dpdt2 = w2 * (EXp_2 - 2.0 * EXp * EXm + EXm_2)
        var_dpdt = w2 * (var_p + var_m)
        var_err = w2 * var_0
        est_i = err2 * var_dpdt
        est_0 = var_err * dpdt2
        if second_order and n0:
            est_i = est_i + var_err * var_dpdt / n0
        EV_i = np.mean(est_i)
        EV_0 = np.mean(est_0)
        mean_condVar = EV_i + EV_0
        if EV_0 < 0:
            dpdt2 = w2 * np.square(EXp - EXm)
            EV_0 = np.mean(var_err * dpdt2)
        if EV_i < 0:
            err2 = np.square(y_eval - w * EX)
            est_i = err2 * var_dpdt
            if second_order and n0:
                est_i = est_i + var_err * var_dpdt / n0
            EV_i = np.mean(est_i)
        if mean_condVar < 0:
            mean_condVar = EV_i + EV_0
        return mean_condVar, EV_i, EV_0

    def _estimate_MSEgrad_Var_condMean(self, y_eval, grad_data_list, size, E0pow_data_list, w=1.0, Scale_para=True,
                                       Eppow_data_list=None, Empow_data_list=None):
        EX_2 = np.array(E0pow_data_list[1])
        if Scale_para:
            EX_3 = np.array(E0pow_data_list[2])
            EX_4 = np.array(E0pow_data_list[3])
            a = -2 * y_eval
            b = 2 * w
            grad_data2_list = a ** 2 * EX_2 + 2.0 * a * b * EX_3 + b ** 2 * EX_4
        else:
            EX = np.array(E0pow_data_list[0])
            EXp = np.array(Eppow_data_list[0])
            EXm = np.array(Empow_data_list[0])
            EXp_2 = np.array(Eppow_data_list[1])
            EXm_2 = np.array(Empow_data_list[1])
            grad_data2_list = (y_eval ** 2 - 2.0 * y_eval * w * EX + w ** 2 * EX_2) * w ** 2 * (EXp_2 + EXm_2 - 2 * EXp * EXm)
        est = np.sum(grad_data2_list) / size - (np.sum(grad_data_list) ** 2 - np.sum(np.square(grad_data_list))) / (size * (size - 1.0))
        if est < 0:
            est = np.var(grad_data_list, ddof=1)
        return est

    def _direct_sample_MSEgrad_varUB_whole(self, y_eval, outcomes_whole_no_shift, w=1, Scale_para=False,
                                           outcomes_whole_p=None, outcomes_whole_m=None):
        if Scale_para:
            out1_list, out2_list = zip(*[split_array_equally(o) for o in outcomes_whole_no_shift])
            out1_array = np.concatenate(out1_list)
            out2_array = np.concatenate(out2_list)
            repeat_counts = [len(o) for o in out1_list]
            grad_outcomes_data = 2 * (w * out1_array - np.repeat(y_eval, repeat_counts)) * out2_array
        else:
            outcomes_0 = np.concatenate(outcomes_whole_no_shift)
            outcomes_p = np.concatenate(outcomes_whole_p)
            outcomes_m = np.concatenate(outcomes_whole_m)
            repeat_counts = [len(o) for o in outcomes_whole_no_shift]
            grad_outcomes_data = (w * outcomes_0 - np.repeat(y_eval, repeat_counts)) * w * (outcomes_p - outcomes_m)
        return np.var(grad_outcomes_data, ddof=1)

    # ---- mini-batches (myQNNs_base.py:1318-1344) ----------------------------------------
    def reset_remain_ind(self):
        self.remain_ind = (np.random.permutation(len(self.x_train))).astype(int)

    def _pop_mini_batch_ind(self, mini_batch_size):
        if len(self.remain_ind) < mini_batch_size:
            eval_ind = np.array(self.remain_ind).copy()
            mini_batch_size -= len(eval_ind)
            self.reset_remain_ind()
            self.epoch_counter[0] += 1
            eval_ind = np.concatenate((eval_ind, self.remain_ind[:mini_batch_size])).astype(int)
            self.remain_ind = self.remain_ind[mini_batch_size:]
        else:
            eval_ind = self.remain_ind[:mini_batch_size]
            self.remain_ind = self.remain_ind[mini_batch_size:]
        return eval_ind

    def _batch_setup(self, mini_batch_size, option_min_approx, option_weight):
        min_approx = self.min_approx_shots if option_min_approx is None else option_min_approx
        wt_dist = self._uniform_wt.copy() if option_weight is None else np.array(option_weight).copy()
        eval_ind = self._pop_mini_batch_ind(mini_batch_size)
        sh = self._shard(len(eval_ind))
        if sh is not None:
            eval_ind = sh.bcast_ints(eval_ind)     # the batch is rank 0's, whatever the ranks' generators did
        return min_approx, np.array(wt_dist)[eval_ind], eval_ind, self.y_train[eval_ind]

    def _shard(self, n_items):
        """This rank's part of a batch of n_items when the object has a process group, else None."""
        if getattr(self, "group", None) is None:
            return None
        from .distributed import BatchShard

        return BatchShard.of(self, n_items)

    def _stats(self, E, y, w, sh, **kw):
        """Batch sums of the estimator epilogue for this rank's rows E (see stats.mse_stats), added over the
        ranks of a sharded batch: (sums [P+1, 8] numpy, unshifted estimates of ALL samples [B] numpy or None,
        per-sample gradient terms of this rank's rows or None).  None when the sums cannot be taken where
        E is (the circuit does not compute on a GPU and no other implementation is plugged in)."""
        fn = getattr(self, "_stats_fn", None)
        if fn is None:
            if not hasattr(E, "detach"):
                return None
            from .stats import mse_stats as fn
        if sh is None:
            S, m0, gd = fn(E, y, w, **kw)
            return np.asarray(S), (None if m0 is None else _np(m0)), gd
        S, m0, gd = fn(E, y[sh.lo:sh.hi], w, row_offset=sh.lo, device_sums=True, **kw)
        S, m0 = sh.reduce(S, m0)
        return S, m0, gd

    # ---- parameter-shift oracles -----------------------------------------------------------
    def MSE_grad_eval(self, params, n_shots_list, mini_batch_size=None, option_min_approx=None, option_weight=None):
        """myQNNs_base.py:1346-1395: grad[i] = mean((w E0 - y) w (E+ - E-)), grad[-1] = mean(2(-y E0 + w E0^2_ub))."""
        grad = np.zeros_like(params)
        min_approx, wt_dist, eval_ind, y_eval = self._batch_setup(mini_batch_size, option_min_approx, option_weight)
        if min_approx == -10:
            # no random numbers in this mode: the products and means are taken where the
            # expectation values are (on the GPU with the HIP circuit) and only the
            # gradient and the unshifted values come back
            return self._exact_grad(params, eval_ind, y_eval)
        if getattr(self, "device_statistics", False):
            g = self._grad_on_device(params, n_shots_list, wt_dist, eval_ind, y_eval, min_approx)
            if g is not None:
                return g
        E, Mg = self._psr_maybe_marg(params, eval_ind, n_shots_list, wt_dist, min_approx)
        n_shots_data_list = np.floor(n_shots_list[-1] * wt_dist).astype(int)
        w = params[-1]
        EX0 = self._estimate(E[:, 0], n_shots_data_list, min_approx, order=4, unbiased=True, marg=Mg(0))
        EX0_list = EX0[0]
        self.cache_train_pred = w * EX0_list
        self.cache_y_train_eval = y_eval
        grad[-1] = np.mean(2.0 * (-y_eval * EX0_list + w * EX0[1]))
        for i in range(len(params) - 1):
            n_shots_data_list = np.floor(n_shots_list[i] * wt_dist).astype(int)
            EXp = self._estimate(E[:, 1 + 2 * i], n_shots_data_list, min_approx, order=2, unbiased=True, marg=Mg(1 + 2 * i))
            EXm = self._estimate(E[:, 2 + 2 * i], n_shots_data_list, min_approx, order=2, unbiased=True, marg=Mg(2 + 2 * i))
            grad[i] = np.mean((w * EX0_list - y_eval) * w * (EXp[0] - EXm[0]))
        return grad

    def _psr_maybe_marg(self, params, eval_ind, n_shots_list, wt_dist, min_approx, force=False):
        """(E [B, 2P+1], column -> readout distribution [B, 4] or None) of a sweep: the
        distributions ride along when some estimate of the call is drawn shot by shot."""
        n_min = np.floor(np.min(n_shots_list) * (np.min(wt_dist) if len(wt_dist) else 1.0))
        if (force or self._may_sample(n_min, min_approx)) and (self.shot_draws == "per_shot" or self._sample_from_marginal):
            E, M = self._psr_all(params, eval_ind, with_marginal=True)
            return E, lambda k: M[:, k]
        return self._psr_all(params, eval_ind), lambda k: None

    def _exact_grad(self, params, eval_ind, y_eval):
        """MSE_grad_eval in exact mode (min_approx_shots = -10; myQNNs_base.py:874-882 makes every
        estimate the exact value), with the same counters."""
        from .distributed import mse_gradient_sums

        from .stats import GD

        B, K = len(eval_ind), 2 * self.n_c_params + 1
        sh = self._shard(B)
        rows = slice(None) if sh is None else slice(sh.lo, sh.hi)
        E = self._psr_all(params, eval_ind[rows], keep_on_device=True)
        w = float(params[-1])
        out = self._stats(E, np.asarray(y_eval, dtype=float), w, sh, want_m0=True)
        if out is not None:
            grad, e0 = out[0][:, GD] / B, out[1]
        else:
            # (a circuit that computes on the host: the same sums in NumPy)
            sums, e0 = mse_gradient_sums(E, np.asarray(y_eval, dtype=float)[rows], w), E[:, 0]
            if sh is not None:
                sums, e0 = sh.reduce(sums, e0)
            grad = sums / B
        self.circuit_counter[0] += B * K
        self.call_counter[0] += B * K
        self.cache_train_pred = w * e0
        self.cache_y_train_eval = y_eval
        out = np.zeros_like(params)
        out[:] = grad
        return out

    def _grad_on_device(self, params, n_shots_list, wt_dist, eval_ind, y_eval, min_approx):
        """MSE_grad_eval with the per-instance statistics taken where the expectation values are:
        either every estimate in the CLT branch (n >= min_approx_shots >= 0, myQNNs_base.py:892-905:
        normal deviates from a torch generator seeded from NumPy's) or every estimate sampled
        (min_approx_shots < 0 or n below it, myQNNs_base.py:715-738: binomial outcome counts from the
        library's device sampler, seeded from NumPy's).  Same estimators, counters and shot totals as
        the host path; the random streams differ from it.  None when the circuit does not compute on
        a GPU or the batch mixes the two branches."""
        n_all = np.floor(np.asarray(n_shots_list, dtype=float)[:, None] * wt_dist[None, :])   # [P+1, B]
        if n_all.min() < 2:
            return None
        if min_approx >= 0 and n_all.min() >= min_approx:
            sampled = False
        elif min_approx < 0 or n_all.max() < min_approx:
            sampled = True
        else:
            return None
        from .stats import CLT, GD, SHOTS

        B, K = len(eval_ind), 2 * self.n_c_params + 1
        sh = self._shard(B)
        rows = slice(None) if sh is None else slice(sh.lo, sh.hi)
        if getattr(self, "_stats_fn", None) is None and not hasattr(self.circuit, "handle"):
            return None
        E = self._psr_all(params, eval_ind[rows], keep_on_device=True)
        P = self.n_c_params
        seed = int(np.random.randint(0, 2 ** 31 - 1))
        if sh is not None:
            seed = int(sh.bcast_ints([seed])[0])
        w = float(params[-1])
        n_list = np.asarray(n_shots_list, dtype=float)
        out = self._stats(E, np.asarray(y_eval, dtype=float), w, sh, shots=n_list, weight=wt_dist[rows], n_shots0=n_list[-1],
                          mode=SHOTS if sampled else CLT, seed=seed, want_m0=True)
        if out is None:
            return None
        sums, m0, _ = out
        self.circuit_counter[0] += B * K
        self.call_counter[0] += (2 if sampled else 1) * B * K           # a sampled state counts twice (myQNNs_base.py:722, 1033)
        self.shots_counter[0] += int(n_all[-1].sum() + 2 * n_all[:P].sum())
        self.cache_train_pred = w * m0
        self.cache_y_train_eval = y_eval
        out = np.zeros_like(params)
        out[:] = sums[:, GD] / B
        return out

    def _ieval_on_device(self, params, n_shots_list, wt_dist, eval_ind, y_eval, min_approx, size, Max_n_shots0,
                         second_order, _estimates=None):
        """MSE_iEval (myQNNs_base.py:1421-1596, the branches without the -10 debug mode) with every
        per-sample array kept on the GPU: one kernel pass over the sweep's matrix (vqa_mse_stats_dev)
        draws the estimates, evaluates the estimator polynomials -- including the alternatives the
        reference falls back to when a mean comes out negative (1242-1248, 1274-1286, 1314-1315) --
        and sums them over the batch; the P+1 rows of sums are finished here ('grad_data_list' stays a
        device tensor).  The reference hands the PARTLY filled (P+1, B) matrix of per-sample
        gradients to its batch-variance estimator (1488, 1578): row sums are accumulated in that
        order here.  None when the circuit does not compute on a GPU or the batch mixes CLT and
        sampled estimates.  _estimates (tests): the [B, 2P+1] matrix of estimates to use."""
        P = self.n_c_params
        n_list = np.asarray(n_shots_list, dtype=float)
        n0 = float(n_list[-1])
        n_shots0 = float(np.max(n_list)) if Max_n_shots0 else n0
        nb_h = np.floor(n_shots0 * wt_dist)                                    # [B]
        ni_h = np.floor(n_list[:P, None] * wt_dist[None, :])                   # [P, B]
        lo = min(nb_h.min(), ni_h.min()) if P else nb_h.min()
        hi = max(nb_h.max(), ni_h.max()) if P else nb_h.max()
        if lo < 4:
            return None
        if min_approx >= 0 and lo >= min_approx:
            sampled = False
        elif min_approx < 0 or hi < min_approx:
            sampled = True
        else:
            return None
        from .stats import ALT_0, ALT_I, CLT, EV_0, EV_I, GD, GD2, GD_SQ, GIVEN, SHOTS

        B, K = len(eval_ind), 2 * P + 1
        sh = self._shard(B)
        rows = slice(None) if sh is None else slice(sh.lo, sh.hi)
        if getattr(self, "_stats_fn", None) is None and not hasattr(self.circuit, "handle"):
            return None
        E = self._psr_all(params, eval_ind[rows], keep_on_device=True)
        w = float(params[-1])
        seed = int(np.random.randint(0, 2 ** 31 - 1))
        if sh is not None:
            seed = int(sh.bcast_ints([seed])[0])
        mode = GIVEN if _estimates is not None else SHOTS if sampled else CLT
        out = self._stats(E, np.asarray(y_eval, dtype=float), w, sh, shots=n_list, weight=wt_dist[rows], n_shots0=n_shots0,
                          mode=mode, seed=seed, want_variances=True, n0=n0, second_order=bool(second_order and n0),
                          estimates=None if _estimates is None else _estimates[rows], want_m0=True, want_grad_data=True)
        if out is None:
            return None
        # (with a process group the per-sample gradient terms 'grad_data_list' stay this rank's rows of the batch;
        # the reference's optimizers drop the entry, vqa/VQA_optimizers/qml_optimizers.py:478)
        S, m0, gd = out
        N_all = float((P + 1) * B)

        def batch_var(sum_gd2, S1, S2):
            """_estimate_MSEgrad_Var_condMean on sums: S1, S2 = sum and sum of squares of the matrix so far."""
            v = sum_gd2 / size - (S1 ** 2 - S2) / (size * (size - 1.0))
            alt = (S2 - S1 ** 2 / N_all) / (N_all - 1.0)                        # np.var(matrix, ddof=1), zero rows included
            return np.where(v < 0, alt, v)

        grad = np.zeros_like(params, dtype=float)
        gvar_b, gvar_EV, gvar_EVi, gvar_EV0 = (np.zeros_like(grad) for _ in range(4))
        grad[:] = S[:, GD] / B
        # scale parameter: conditional variance (or its fallback) and the batch variance
        ev_last = S[P, EV_I] / B
        if ev_last < 0:
            ev_last = S[P, ALT_I] / B
        gvar_EV[-1] = gvar_EV0[-1] = ev_last
        gvar_b[-1] = float(batch_var(S[P, GD2], S[P, GD], S[P, GD_SQ]))
        if P:
            ev_i, ev_0 = S[:P, EV_I] / B, S[:P, EV_0] / B
            mcv = ev_i + ev_0
            ev_0 = np.where(ev_0 < 0, S[:P, ALT_0] / B, ev_0)
            ev_i = np.where(ev_i < 0, S[:P, ALT_I] / B, ev_i)
            mcv = np.where(mcv < 0, ev_i + ev_0, mcv)
            S1 = S[P, GD] + np.cumsum(S[:P, GD])
            S2 = S[P, GD_SQ] + np.cumsum(S[:P, GD_SQ])
            gvar_EV[:P], gvar_EVi[:P], gvar_EV0[:P] = mcv, ev_i, ev_0
            gvar_b[:P] = batch_var(S[:P, GD2], S1, S2)
        self.circuit_counter[0] += B * K
        self.call_counter[0] += (2 if sampled else 1) * B * K
        self.shots_counter[0] += int(nb_h.sum() + 2 * ni_h.sum())
        self.cache_train_pred = w * m0
        self.cache_y_train_eval = y_eval
        return {"grad": grad, "gvar_list_EV": gvar_EV, "gvar_list_EVi": gvar_EVi, "gvar_list_EV0": gvar_EV0,
                "gvar_list_b": gvar_b, "grad_data_list": gd}

    def MSE_loss_eval(self, params, n_shots, mini_batch_size=None, option_min_approx=None, option_weight=None):
        """myQNNs_base.py:1397-1419."""
        min_approx, wt_dist, eval_ind, y_eval = self._batch_setup(mini_batch_size, option_min_approx, option_weight)
        n_shots_data_list = np.floor(n_shots * wt_dist).astype(int)
        exact, marg = self._expect_maybe_marg(params, n_shots_data_list, min_approx, which="train", idx=eval_ind)
        EX0 = self._estimate(exact, n_shots_data_list, min_approx, order=2, unbiased=True, marg=marg)
        w = params[-1]
        return np.mean(w ** 2 * EX0[1] - 2.0 * w * y_eval * EX0[0] + np.square(y_eval))

    def MSE_iEval(self, params, n_shots_list, option_min_approx=None, option_weight=None, mini_batch_size=None,
                  Max_n_shots0=True, second_order=True):
        """myQNNs_base.py:1421-1596: gradient plus the variance estimates the shot-adaptive optimizers consume."""
        grad = np.zeros_like(params)
        gvar_list_b = np.zeros_like(params)
        gvar_list_EV = np.zeros_like(params)
        gvar_list_EVi = np.zeros_like(params)
        gvar_list_EV0 = np.zeros_like(params)
        min_approx, wt_dist, eval_ind, y_eval = self._batch_setup(mini_batch_size, option_min_approx, option_weight)
        size = mini_batch_size
        if getattr(self, "device_statistics", False) and min_approx != -10:
            out = self._ieval_on_device(params, n_shots_list, wt_dist, eval_ind, y_eval, min_approx, size,
                                        Max_n_shots0, second_order)
            if out is not None:
                return out
        grad_data_list = np.zeros((len(params), len(eval_ind)))
        debug = min_approx == -10
        E, Mg = self._psr_maybe_marg(params, eval_ind, n_shots_list, wt_dist, min_approx, force=debug)
        n0 = n_shots_list[-1]
        n_shots0 = np.max(n_shots_list) if Max_n_shots0 else n_shots_list[-1]
        n_shots_data_list = np.floor(n_shots0 * wt_dist).astype(int)
        w = params[-1]
        if debug:
            EX0pow_exact, var_0_exact = self._estimate(E[:, 0], n_shots_data_list, -10, order=4, Return_var=True, unbiased=True)
            EX0pow, var_0 = self._estimate(E[:, 0], n_shots_data_list, -1, order=4, Return_var=True, unbiased=True, marg=Mg(0))
            grad_data_exact = 2.0 * (-y_eval * EX0pow_exact[0] + w * EX0pow_exact[1])
        else:
            EX0pow, var_0 = self._estimate(E[:, 0], n_shots_data_list, min_approx, order=4, Return_var=True, unbiased=True,
                                           marg=Mg(0))
        EX0_list = EX0pow[0]
        self.cache_train_pred = w * EX0_list
        self.cache_y_train_eval = y_eval
        grad_data_list[-1] = 2.0 * (-y_eval * EX0_list + w * EX0pow[1])
        grad[-1] = np.mean(grad_data_list[-1])
        if debug:
            gvar_list_EV[-1] = self._estimate_MSEgrad_mean_condVar(y_eval, EX0pow_exact, w=w, Scale_para=True,
                                                                   var_0=var_0_exact, n0=n0, second_order=second_order)
            gvar_list_b[-1] = self._estimate_MSEgrad_Var_condMean(y_eval, grad_data_exact, size, EX0pow_exact, w=w, Scale_para=True)
        else:
            gvar_list_EV[-1] = self._estimate_MSEgrad_mean_condVar(y_eval, EX0pow, w=w, Scale_para=True, var_0=var_0,
                                                                   n0=n0, second_order=second_order)
            gvar_list_EV0[-1] = gvar_list_EV[-1]
            # the reference passes the whole (n_params, B) matrix here (myQNNs_base.py:1488)
            gvar_list_b[-1] = self._estimate_MSEgrad_Var_condMean(y_eval, grad_data_list, size, EX0pow, w=w, Scale_para=True)
        for i in range(len(params) - 1):
            n_shots_data_list = np.floor(n_shots_list[i] * wt_dist).astype(int)
            Ep, Em = E[:, 1 + 2 * i], E[:, 2 + 2 * i]
            if debug:
                EXp_exact, var_p_exact = self._estimate(Ep, n_shots_data_list, -10, order=2, Return_var=True, unbiased=True)
                EXm_exact, var_m_exact = self._estimate(Em, n_shots_data_list, -10, order=2, Return_var=True, unbiased=True)
                EXp, var_p = self._estimate(Ep, n_shots_data_list, -1, order=2, Return_var=True, unbiased=True, marg=Mg(1 + 2 * i))
                EXm, var_m = self._estimate(Em, n_shots_data_list, -1, order=2, Return_var=True, unbiased=True, marg=Mg(2 + 2 * i))
                grad_data_exact = (w * EX0pow_exact[0] - y_eval) * w * (EXp_exact[0] - EXm_exact[0])
            else:
                EXp, var_p = self._estimate(Ep, n_shots_data_list, min_approx, order=2, Return_var=True, unbiased=True,
                                            marg=Mg(1 + 2 * i))
                EXm, var_m = self._estimate(Em, n_shots_data_list, min_approx, order=2, Return_var=True, unbiased=True,
                                            marg=Mg(2 + 2 * i))
            grad_data_list[i] = (w * EX0_list - y_eval) * w * (EXp[0] - EXm[0])
            grad[i] = np.mean(grad_data_list[i])
            if debug:
                gvar_list_EV[i] = self._estimate_MSEgrad_mean_condVar(
                    y_eval, EX0pow_exact, w=w, second_order=second_order, n0=n0, var_0=var_0_exact, Scale_para=False,
                    Eppow_data_list=EXp_exact, Empow_data_list=EXm_exact, var_p=var_p_exact, var_m=var_m_exact)[0]
                gvar_list_b[i] = self._estimate_MSEgrad_Var_condMean(y_eval, grad_data_exact, size, EX0pow_exact, w=w,
                                                                     Scale_para=False, Eppow_data_list=EXp_exact,
                                                                     Empow_data_list=EXm_exact)
            else:
                EV, EVi, EV0 = self._estimate_MSEgrad_mean_condVar(
                    y_eval, EX0pow, w=w, second_order=second_order, n0=n0, var_0=var_0, Scale_para=False,
                    Eppow_data_list=EXp, Empow_data_list=EXm, var_p=var_p, var_m=var_m)
                gvar_list_EV[i], gvar_list_EVi[i], gvar_list_EV0[i] = EV, EVi, EV0
                gvar_list_b[i] = self._estimate_MSEgrad_Var_condMean(y_eval, grad_data_list, size, EX0pow, w=w, Scale_para=False,
                                                                     Eppow_data_list=EXp, Empow_data_list=EXm)
        return {"grad": grad, "gvar_list_EV": gvar_list_EV, "gvar_list_EVi": gvar_list_EVi,
                "gvar_list_EV0": gvar_list_EV0, "gvar_list_b": gvar_list_b, "grad_data_list": grad_data_list}

    def _outcomes(self, exact, n_s, marg=None):
        """Raw +-1 outcome arrays (one per sample) drawn from the exact Z0 distribution
        (stands in for _sample_outcomes_sum_Z_all_data, myQNNs_base.py:749-765)."""
        self.circuit_counter[0] += len(exact)
        self.call_counter[0] += len(exact)
        self.shots_counter[0] += int(np.sum(n_s))
        out = []
        for j, (e, n) in enumerate(zip(exact, n_s)):
            if self.shot_draws == "per_shot":
                out.append((-1) ** (self._shots_low2(marg[j], n) % 2))
            else:
                out.append(np.where(np.random.random(int(n)) < 0.5 * (1.0 - e), -1, 1))
        return out, np.array([o.sum() for o in out], dtype=float)

    def MSE_iEval_iRandom(self, params, n_shots_list, option_weight=None):
        """myQNNs_base.py:1619-1666: per component an independent multinomial shot
        allocation over the whole training set (gCANS / Refoqus)."""
        grad = np.zeros_like(params)
        gvar_list = np.zeros_like(params)
        if option_weight is None:
            wt_dist = self._uniform_wt_dist.copy()
        else:
            wt_dist = np.array(option_weight) / np.sum(option_weight)
        w = params[-1]
        M_tot = self.num_train_data
        theta = np.asarray(params[:self.n_c_params], dtype=np.float64)
        per_shot = self.shot_draws == "per_shot"
        if getattr(self, "device_statistics", False) and not per_shot and hasattr(self.circuit, "handle"):
            return self._irandom_on_device(params, n_shots_list, wt_dist, theta)
        for i in range(len(params)):
            n_shots_data_list = np.random.multinomial(n_shots_list[i], wt_dist)
            sel = np.nonzero(n_shots_data_list != 0)[0].astype(np.int32)
            y_eval = self.y_train[sel]
            n_shots_eval = n_shots_data_list[sel]
            n = n_shots_list[i]
            if i == len(params) - 1:
                n_shots_eval = n_shots_eval * 2
                exact, marg = self._expect_all(params, which="train", idx=sel, with_marginal=True) if per_shot \
                    else (self._expect_all(params, which="train", idx=sel), None)
                outcomes0, outsum = self._outcomes(exact, n_shots_eval, marg)
                grad[i] = np.sum(w * (outsum ** 2 - n_shots_eval) / (2 * n * (n - 1) / M_tot + n) - y_eval * outsum / n)
                gvar_list[i] = self._direct_sample_MSEgrad_varUB_whole(y_eval, outcomes0, w=w, Scale_para=True)
            else:
                E = len(sel)
                si = np.tile(sel, 3)
                sp = np.concatenate([np.full(E, -1), np.full(E, i), np.full(E, i)]).astype(np.int32)
                sv = np.concatenate([np.zeros(E), np.full(E, 0.5 * np.pi), np.full(E, -0.5 * np.pi)])
                e, m = self.circuit.eval_batch(theta, self._handle("train"), sample_idx=si, shift_param=sp, shift_val=sv,
                                               obs_list=self._obs_list(), n_marg_bits=self._marg_bits() if per_shot else 0)
                e = _np(e)[:, 0]
                m = self._pad4(m) if per_shot else [None] * (3 * E)
                outcomes0, outsum0 = self._outcomes(e[:E], n_shots_eval, m[:E])
                outcomes_p, outsum_p = self._outcomes(e[E:2 * E], n_shots_eval, m[E:2 * E])
                outcomes_m, outsum_m = self._outcomes(e[2 * E:], n_shots_eval, m[2 * E:])
                grad[i] = np.sum((w * outsum0 / (n + n * (n - 1) / M_tot) - y_eval / n) * (outsum_p - outsum_m) * w)
                gvar_list[i] = self._direct_sample_MSEgrad_varUB_whole(y_eval, outcomes0, w=w, Scale_para=False,
                                                                       outcomes_whole_p=outcomes_p, outcomes_whole_m=outcomes_m)
        return {"grad": grad, "gvar_list": gvar_list}


    def _irandom_on_device(self, params, n_shots_list, wt_dist, theta, kind=0):
        """MSE_iEval_iRandom with everything behind the shot allocation on the GPU: the multinomial
        allocations of all components are drawn here (NumPy's generator, the reference's order,
        myQNNs_base.py:1637), the (component, sample) rows that received shots become ONE explicit
        instance list, and a kernel draws the joint outcome counts of every row and reduces them to
        the gradient and variance estimates (vqa_irandom_stats_dev).  Same estimators, counters and
        shot totals as the host path; the outcome stream is the library's."""
        from .stats import irandom_stats

        C_ = len(params)
        has_scale = kind == 0
        rows, shots, seg = [], [], [0]
        for i in range(C_):
            alloc = np.random.multinomial(n_shots_list[i], wt_dist)
            sel = np.nonzero(alloc != 0)[0]
            rows.append(sel)
            shots.append(alloc[sel])
            seg.append(seg[-1] + len(sel))
        rows = np.concatenate(rows).astype(np.int32)
        shots = np.concatenate(shots).astype(np.int64)
        seed = int(np.random.randint(0, 2 ** 31 - 1))
        if kind == 0:
            out = irandom_stats(self.circuit, theta, self._handle("train"), self._obs_list(), rows, seg, shots,
                                np.asarray(n_shots_list, dtype=float), float(params[-1]), self.num_train_data, seed,
                                kind=0, has_scale=True, rows_y=self.y_train[rows])
        else:
            out = irandom_stats(self.circuit, theta, self._handle("train"), self.class_observables(), rows, seg, shots,
                                np.asarray(n_shots_list, dtype=float), 1.0, self.num_train_data, seed,
                                kind=1, has_scale=False, rows_label=np.asarray(self.y_train, dtype=np.int64)[rows])
        n_circ_rows = seg[-2] if has_scale else seg[-1]
        n_scale_rows = seg[-1] - n_circ_rows
        self.circuit_counter[0] += 3 * n_circ_rows + n_scale_rows
        self.call_counter[0] += 3 * n_circ_rows + n_scale_rows
        self.shots_counter[0] += int(3 * shots[:n_circ_rows].sum() + 2 * shots[n_circ_rows:].sum())
        return {"grad": out[:, 0].copy(), "gvar_list": out[:, 1].copy()}


class myQNN_regression(myQNN_base):
    """vqa/QNN/myQNNs_regression.py:256: the regressor kept as a standalone class there.  Same
    circuit path as myQNN_base; its gradient oracles add an L2 penalty that pulls the circuit
    parameters to 0 and the scale parameter to `reg_w_coef * max|y|` (myQNNs_regression.py:
    1311-1320, 1352-1355, 1560-1564, 1644-1647), and MSE_iEval defaults to first-order variance
    terms (second_order=False, myQNNs_regression.py:1381)."""

    def set_data(self, x_train, y_train, x_test, y_test):
        super().set_data(x_train, y_train, x_test, y_test)
        if x_train is not None and y_train is not None:
            self._max_abs_y_train = np.max(np.abs(self.y_train))     # myQNNs_regression.py:342

    def _l2_centre(self, params, y_eval, Online_reg_w_center, reg_w_coef):
        """Centre of the penalty: zero for the circuit parameters, reg_w_coef * (largest label
        seen so far | largest |label| of the training set) for the scale (1311-1320)."""
        if Online_reg_w_center:
            # the reference tracks max(y), not max|y|, in the online variant (1312-1315)
            cur = np.max(y_eval)
            self._max_abs_y_online = cur if not hasattr(self, "_max_abs_y_online") else max(cur, self._max_abs_y_online)
            w_center = self._max_abs_y_online
        else:
            w_center = self._max_abs_y_train
        cent = np.zeros_like(params)
        cent[-1] = w_center * reg_w_coef
        return cent

    @staticmethod
    def _l2_term(params, L2_reg_lmd, cent, Take_abs):
        sign_for_w = np.full_like(params, 1.0)
        if Take_abs:
            sign_for_w[-1] = np.sign(params[-1])          # penalty (|w| - a)^2 on the scale
        return L2_reg_lmd * (params - sign_for_w * cent)

    def MSE_grad_eval(self, params, n_shots_list, mini_batch_size=None, option_min_approx=None, option_weight=None,
                      L2_reg_lmd=0.0, Online_reg_w_center=False, reg_w_coef=1, Take_abs=False):
        """myQNNs_regression.py:1293-1355."""
        params = np.asarray(params, dtype=float)
        grad = super().MSE_grad_eval(params, n_shots_list, mini_batch_size=mini_batch_size,
                                     option_min_approx=option_min_approx, option_weight=option_weight)
        cent = self._l2_centre(params, self.cache_y_train_eval, Online_reg_w_center, reg_w_coef)
        return grad + self._l2_term(params, L2_reg_lmd, cent, Take_abs)

    def MSE_iEval(self, params, n_shots_list, option_min_approx=None, option_weight=None, mini_batch_size=None,
                  Max_n_shots0=True, second_order=False, L2_reg_lmd=0.0, Online_reg_w_center=False, reg_w_coef=1.5,
                  WoR_correct=False, Take_abs=False):
        """myQNNs_regression.py:1381-1571 (WoR_correct is accepted and unused there as well)."""
        params = np.asarray(params, dtype=float)
        out = super().MSE_iEval(params, n_shots_list, option_min_approx=option_min_approx, option_weight=option_weight,
                                mini_batch_size=mini_batch_size, Max_n_shots0=Max_n_shots0, second_order=second_order)
        cent = self._l2_centre(params, self.cache_y_train_eval, Online_reg_w_center, reg_w_coef)
        out["grad"] = out["grad"] + self._l2_term(params, L2_reg_lmd, cent, Take_abs)
        return out

    def MSE_iEval_iRandom(self, params, n_shots_list, option_weight=None, L2_reg_lmd=0.0, L2_reg_cent=None,
                          reg_w_coef=1.5, Take_abs=False):
        """myQNNs_regression.py:1594-1648 (the L2_reg_cent argument is overwritten there, 1609-1611)."""
        params = np.asarray(params, dtype=float)
        out = super().MSE_iEval_iRandom(params, n_shots_list, option_weight=option_weight)
        cent = np.zeros_like(params)
        cent[-1] = self._max_abs_y_train * reg_w_coef
        out["grad"] = out["grad"] + self._l2_term(params, L2_reg_lmd, cent, Take_abs)
        return out
