"""3-class QNN classifier target object on the batched HIP engine.

Counterpart of ``myQNN_classifier`` (vqa/QNN/myQNNs_classifier.py:48-1285) for
noiseless circuits: class probabilities are read from qubits 0 and 1, either
through the unbiased three-outcome POVM (outcome 3 is handed to a uniformly
random class; ``class_unbias=True``) or as ``(s % 4) % 3``.

The reference evaluates one circuit per sample, shift and parameter and draws
shots from each state (myQNNs_classifier.py:600-665, 844-859).  Here one
batched sweep returns, for every instance, the exact distribution of the two
readout qubits (a 4-entry marginal), and the shot statistics are drawn from
those with NumPy in the same call order: unshifted batch first, then per
parameter the +shift batch and the -shift batch.  The reference draws basis
states with qulacs' private generator, so finite-shot results agree in
distribution, not draw by draw; exact quantities agree to rounding.

Zero-noise extrapolation (``use_QEM``) needs the reference's noisy circuit
objects and is not offered.
"""
from __future__ import annotations

import numpy as np

from .observable import Observable
from .qnn_base import myQNN_base, _np


def label2one_hot(label, num_classes):
    """vqa/QNN/myQNNs_base.py:223-226."""
    v = np.zeros(num_classes)
    v[label] = 1
    return v


class myQNN_classifier(myQNN_base):
    def __init__(self, n_qubits, learning_circuit, x_train=None, y_train=None, x_test=None, y_test=None,
                 min_approx_shots=-1, setting_dict=None, id_string="QNN", file_id=None, obs=None, num_classes=3,
                 shot_time=1.0, per_circuit_overhead=0.0, communication_overhead=0.0,
                 max_sub_nqubits_sampling_aft_partial_tr=6, partial_MSE=False, eps=0.0, class_unbias=True,
                 noisy=False, device_statistics=False, shot_draws="binomial", **unused):
        if noisy:
            raise NotImplementedError("noisy circuits / error mitigation are outside this backend")
        if num_classes != 3:
            raise NotImplementedError("the readout POVM is defined for 3 classes (as in the reference)")
        self.partial_MSE = partial_MSE
        self.num_classes = num_classes
        self.class_unbias = class_unbias
        self._eps = eps
        self._class_obs = None
        super().__init__(n_qubits, learning_circuit, x_train, y_train, x_test, y_test, min_approx_shots, setting_dict,
                         id_string, file_id, None, shot_time, per_circuit_overhead, communication_overhead,
                         max_sub_nqubits_sampling_aft_partial_tr, device_statistics=device_statistics,
                         shot_draws=shot_draws)
        self.n_params = self.n_c_params  # no re-normalisation coefficient

    # ---- data ---------------------------------------------------------------------
    def set_data(self, x_train, y_train, x_test, y_test):
        for y in (y_train, y_test):
            if y is not None and int(np.max(y)) + 1 > self.num_classes:
                raise NotImplementedError("labels beyond class 2: the readout POVM is defined for 3 classes")
        self.num_bit = len(bin(self.num_classes - 1)) - 2
        mod2n = 2 ** self.num_bit
        self.interpret = lambda x: (x % mod2n) % self.num_classes
        super().set_data(x_train, y_train, x_test, y_test)
        if x_train is not None and y_train is not None:
            self.y_train_1hot = np.array([label2one_hot(int(v), self.num_classes) for v in y_train])
        if x_test is not None and y_test is not None:
            self.y_test_1hot = np.array([label2one_hot(int(v), self.num_classes) for v in y_test])

    def __getstate__(self):
        d = super().__getstate__()
        d.pop("interpret", None)
        d["_class_obs"] = None
        d["_z0"] = None
        return d

    def __setstate__(self, d):
        super().__setstate__(d)
        mod2n = 2 ** self.num_bit
        self.interpret = lambda x: (x % mod2n) % self.num_classes

    _LOSS_METHODS = {
        "mean_squared_error": {
            "func_to_track": "exact_predict_error_and_MSE", "loss": "MSE_loss_eval", "grad": "MSE_grad_eval",
            "iEvaluate": "MSE_iEval", "iEvaluate_iRandom": "MSE_iEval_iRandom",
        },
        "cross_entropy": {
            "func_to_track": "exact_error_both_to_track", "loss": "CE_loss_eval", "grad": "CE_grad_eval",
            "iEvaluate": "CE_iEval", "iEvaluate_iRandom": "CE_iEval_iRandom",
        },
    }

    def compile(self, *args, partial_MSE=False, eps=0.0, **kwargs):
        self.partial_MSE = partial_MSE
        self._eps = eps
        super().compile(*args, **kwargs)

    # the inherited Z0 estimators (MSE_loss_eval) evaluate this class's default observable
    # (Z0 Z1, myQNNs_classifier.py:282-293) in their exact / CLT branches but read qubit 0 of the
    # drawn basis states in their finite-shot branch (myQNNs_classifier.py:549-572)
    _sample_from_marginal = True

    def _default_obs(self):
        """myQNNs_classifier.py:282-293."""
        obs = Observable(self.n_qubits)
        obs.add_operator(1.0, " ".join(f"Z {i}" for i in range(self.num_bit)))
        return obs

    # ---- exact class probabilities (myQNNs_classifier.py:295-365) ----------------------
    def class_observables(self):
        if self._class_obs is None:
            if self.class_unbias:
                spec = [
                    [(1 / 3, ""), (1 / 3, "Z 0 Z 1"), (1 / 6, "Z 0"), (1 / 6, "Z 1")],
                    [(1 / 3, ""), (-1 / 3, "Z 0"), (1 / 6, "Z 1"), (-1 / 6, "Z 0 Z 1")],
                    [(1 / 3, ""), (-1 / 3, "Z 1"), (1 / 6, "Z 0"), (-1 / 6, "Z 0 Z 1")],
                ]
            else:
                spec = [
                    [(0.5, ""), (0.5, "Z 0 Z 1")],
                    [(0.25, ""), (-0.25, "Z 0"), (0.25, "Z 1"), (-0.25, "Z 0 Z 1")],
                    [(0.25, ""), (0.25, "Z 0"), (-0.25, "Z 1"), (-0.25, "Z 0 Z 1")],
                ]
            self._class_obs = []
            for terms in spec:
                o = Observable(self.n_qubits)
                for c, s in terms:
                    o.add_operator(c, s)
                self._class_obs.append(o)
        return self._class_obs

    def _prob_from_marginal(self, m):
        """Class probabilities [.., 3] from the distribution of (q0 + 2 q1) [.., 4]."""
        m = np.asarray(m)
        if self.class_unbias:
            return m[..., :3] + m[..., 3:4] / 3.0
        return np.stack([m[..., 0] + m[..., 3], m[..., 1], m[..., 2]], axis=-1)

    def _exact_probs(self, params, x_list=None, which=None, idx=None):
        theta = np.asarray(params[:self.n_c_params], dtype=np.float64)
        if x_list is not None:
            inputs = self.circuit.upload_inputs(np.asarray(x_list))
            e, _ = self.circuit.eval_batch(theta, inputs, obs_list=self.class_observables())
        else:
            e, _ = self.circuit.eval_batch(theta, self._handle(which), sample_idx=None if idx is None else np.asarray(idx, np.int32),
                                           obs_list=self.class_observables())
        return _np(e)

    def exact_prob_dist_data_by_data(self, x_list, params):
        return self._exact_probs(params, x_list=x_list)

    def _y_pred_w_err(self, prob_dist_data_list, eps=0.0):
        """myQNNs_classifier.py:440-451."""
        sorted_indices = np.argsort(prob_dist_data_list, axis=1)[:, -2:]
        top_values = np.take_along_axis(prob_dist_data_list, sorted_indices, axis=1)
        diff = top_values[:, -1] - top_values[:, -2]
        return np.where(diff > 2.0 * eps, sorted_indices[:, -1], -1)

    def _predict_from_probs(self, p, eps):
        """myQNNs_classifier.py:367-387: the last class is 1 - sum of the others."""
        p = np.array(p, dtype=float)
        p[:, -1] = 1.0 - p[:, :-1].sum(axis=1)
        i_max = np.argmax(p, axis=1)
        top2 = np.sort(p, axis=1)[:, -2:]
        return np.where(top2[:, 1] - top2[:, 0] < 2 * eps, -1, i_max)

    def exact_predict(self, x_list, params=None, eps=0.0):
        if params is None:
            params = self.model_params
        return self._predict_from_probs(self._exact_probs(params, x_list=x_list), eps)

    def exact_error_rate(self, params, y_pred_dict=None, which="test", ret_both=False, log_evol=False, eps=0.0):
        if ret_both:
            which = "test"
        if which == "train":
            y_pred = (self._predict_from_probs(self._exact_probs(params, which="train"), eps).astype(float)
                      if y_pred_dict is None else y_pred_dict["train"])
            error = 1 - np.mean(np.asarray(self.y_train) == np.asarray(y_pred))
            if log_evol:
                self.exact_cost_evol.append(self.exact_error_rate(params, which="test"))
            return error
        y_pred = (self._predict_from_probs(self._exact_probs(params, which="test"), eps).astype(float)
                  if y_pred_dict is None else y_pred_dict["test"])
        error = 1 - np.mean(np.asarray(self.y_test) == np.asarray(y_pred))
        if log_evol:
            self.exact_cost_evol.append(error)
        if ret_both:
            return {"test": error, "train": self.exact_error_rate(params, which="train", ret_both=False)}
        return error

    def exact_error_both_to_track(self, params, y_pred_dict=None, eps=0.0):
        return self.exact_error_rate(params, y_pred_dict=y_pred_dict, which="test", ret_both=True, log_evol=False, eps=eps)

    def exact_MSE(self, params, which="test", ret_both=False, log_evol=False, partial_MSE=False, ret_prediction=False, eps=0.0):
        """myQNNs_classifier.py:453-492."""
        if ret_both:
            which = "test"
        if which == "train":
            p = self._exact_probs(params, which="train").astype(float)
            y_pred = self._y_pred_w_err(p, eps=eps) if ret_prediction else None
            if partial_MSE:
                p = p * self.y_train_1hot
            MSE = np.sum(np.square(p - self.y_train_1hot)) / len(self.y_train_1hot)
            if log_evol:
                self.exact_cost_evol.append(self.exact_MSE(params, which="test"))
            return (MSE, y_pred) if ret_prediction else MSE
        p = self._exact_probs(params, which="test").astype(float)
        y_pred_test = self._y_pred_w_err(p, eps=eps) if ret_prediction else None
        if partial_MSE:
            p = p * self.y_test_1hot
        MSE = np.sum(np.square(p - self.y_test_1hot)) / len(self.y_test_1hot)
        if log_evol:
            self.exact_cost_evol.append(MSE)
        if ret_both:
            if ret_prediction:
                MSE_train, y_pred_train = self.exact_MSE(params, which="train", ret_prediction=True, partial_MSE=partial_MSE, eps=eps)
                return [MSE, MSE_train], {"train": y_pred_train, "test": y_pred_test}
            return [MSE, self.exact_MSE(params, which="train", ret_prediction=False)]
        return (MSE, y_pred_test) if ret_prediction else MSE

    def exact_MAE(self, params, which="test", ret_both=False, log_evol=False, partial_MSE=False, ret_prediction=False):
        """myQNNs_classifier.py:494-533: despite its name the same squared error as exact_MSE, with
        plain argmax predictions; the train half of ret_both comes from exact_MSE (523-527)."""
        if ret_both:
            which = "test"
        if which == "train":
            p = self._exact_probs(params, which="train").astype(float)
            y_pred = np.argmax(p, axis=1) if ret_prediction else None
            if partial_MSE:
                p = p * self.y_train_1hot
            MSE = np.sum(np.square(p - self.y_train_1hot)) / len(self.y_train_1hot)
            if log_evol:
                self.exact_cost_evol.append(self.exact_MSE(params, which="test"))
            return (MSE, y_pred) if ret_prediction else MSE
        p = self._exact_probs(params, which="test").astype(float)
        y_pred_test = np.argmax(p, axis=1) if ret_prediction else None
        if partial_MSE:
            p = p * self.y_test_1hot
        MSE = np.sum(np.square(p - self.y_test_1hot)) / len(self.y_test_1hot)
        if log_evol:
            self.exact_cost_evol.append(MSE)
        if ret_both:
            if ret_prediction:
                MSE_train, y_pred_train = self.exact_MSE(params, which="train", ret_prediction=True)
                return [MSE, MSE_train], {"train": y_pred_train, "test": y_pred_test}
            return [MSE, self.exact_MSE(params, which="train", ret_prediction=False)]
        return (MSE, y_pred_test) if ret_prediction else MSE

    def exact_predict_error_and_MSE(self, params, loss="MSE", eps=None):
        """myQNNs_classifier.py:536-547."""
        if eps is None:
            eps = self._eps
        MSE_list, y_pred_dict = self.exact_MSE(params, which="test", ret_both=True, log_evol=False,
                                               partial_MSE=self.partial_MSE, ret_prediction=True, eps=eps)
        err_dict = self.exact_error_both_to_track(params, y_pred_dict=y_pred_dict)
        err_dict.update({"test loss": MSE_list[0], "train loss": MSE_list[1]})
        return err_dict

    def generate_labels(self, x_data, target_params):
        return self.exact_predict(x_data, target_params)

    # ---- batched sweeps and shot statistics ------------------------------------------------
    def _psr_marginals(self, params, idx):
        """Distribution of the two readout qubits for the 2P+1 shift instances of each selected sample: [B, 2P+1, 4]."""
        theta = np.asarray(params[:self.n_c_params], dtype=np.float64)
        _, m = self.circuit.psr_full(theta, self._handle("train"), sel=np.asarray(idx, np.int32), obs_list=[], n_marg_bits=2)
        return _np(m)

    def _draw_class_counts(self, marg, n_s):
        """Shot counts per class [B, 3] (myQNNs_classifier.py:607-622): multinomial over the four
        readout outcomes as a chain of binomials, outcome 3 re-dealt uniformly when class_unbias."""
        marg = np.clip(np.asarray(marg, dtype=float), 0.0, 1.0)
        n_s = np.asarray(n_s).astype(np.int64)
        B = len(n_s)
        self.circuit_counter[0] += B
        self.call_counter[0] += B
        self.shots_counter[0] += int(n_s.sum())
        if self.shot_draws == "per_shot":
            return np.array([np.bincount(self._class_outcomes(marg[j], n_s[j]), minlength=3)[:3] for j in range(B)],
                            dtype=float).reshape(B, 3)
        k = np.zeros((B, 4), dtype=np.int64)
        left = n_s.copy()
        rest = np.ones(B)
        for o in range(3):
            with np.errstate(divide="ignore", invalid="ignore"):
                q = np.where(rest > 0, marg[:, o] / rest, 0.0)
            k[:, o] = np.random.binomial(left, np.clip(q, 0.0, 1.0))
            left = left - k[:, o]
            rest = rest - marg[:, o]
        k[:, 3] = left
        if self.class_unbias:
            a = np.random.binomial(k[:, 3], 1.0 / 3.0)
            b = np.random.binomial(k[:, 3] - a, 0.5)
            return np.stack([k[:, 0] + a, k[:, 1] + b, k[:, 2] + (k[:, 3] - a - b)], axis=1).astype(float)
        return np.stack([k[:, 0] + k[:, 3], k[:, 1], k[:, 2]], axis=1).astype(float)

    def _class_outcomes(self, marg_row, n):
        """Class index of each of n shots of one state, drawn one by one
        (myQNNs_classifier.py:613-619, 640-647): the readout of qubits (0, 1), then outcome 3 dealt
        to a uniformly random class (class_unbias) or folded onto class 0."""
        outcomes = self._shots_low2(marg_row, n)
        if self.class_unbias:
            idx3 = np.where(outcomes == 3)[0]
            outcomes[idx3] = np.random.choice([0, 1, 2], size=len(idx3))
            return outcomes
        return outcomes % 3

    @staticmethod
    def _count_stats(count, n):
        """p, unbiased p^2, sample variance of the 0/1 outcomes (myQNNs_classifier.py:623-627, 661-665)."""
        n = np.asarray(n, dtype=float)
        if count.ndim == 2:
            n = n[:, None]
        with np.errstate(divide="ignore", invalid="ignore"):
            p = count / n
            varp = count / (n - 1.0) - np.square(count) / (n * (n - 1.0))
            p2 = (n / (n - 1.0)) * np.square(p) - p / (n - 1.0)
        return p, p2, varp

    def _success(self, counts, y_eval):
        return counts[np.arange(len(y_eval)), np.asarray(y_eval, dtype=int)]

    def _setup(self, mini_batch_size, option_weight):
        wt_dist = self._uniform_wt.copy() if option_weight is None else np.array(option_weight).copy()
        eval_ind = self._pop_mini_batch_ind(mini_batch_size)
        sh = self._shard(len(eval_ind))
        if sh is not None:
            eval_ind = sh.bcast_ints(eval_ind)     # the batch is rank 0's
        return np.array(wt_dist)[eval_ind], eval_ind, self.y_train[eval_ind], self.y_train_1hot[eval_ind]

    def MSE_grad_eval(self, params, n_shots_list, option_min_approx=None, option_weight=None, mini_batch_size=None,
                      Max_n_shots0=True, second_order=True, partial_MSE=True, use_QEM=False, **unused):
        """myQNNs_classifier.py:1055-1156 without error mitigation."""
        if use_QEM:
            raise NotImplementedError("error mitigation is outside this backend")
        self.partial_MSE = partial_MSE
        grad = np.zeros_like(params)
        wt_dist, eval_ind, y_eval, y_1hot = self._setup(mini_batch_size, option_weight)
        if getattr(self, "device_statistics", False):
            g = self._grad_on_device(params, n_shots_list, wt_dist, eval_ind, y_eval, y_1hot, partial_MSE)
            if g is not None:
                return g
        M = self._psr_marginals(params, eval_ind)
        n0 = np.floor(np.max(n_shots_list) * wt_dist).astype(int)
        c0 = self._draw_class_counts(M[:, 0], n0)
        if partial_MSE:
            p0 = self._success(c0, y_eval) / n0
        else:
            err = c0 / n0[:, None] - y_1hot
        for i in range(len(params)):
            n_i = np.floor(n_shots_list[i] * wt_dist).astype(int)
            cp = self._draw_class_counts(M[:, 1 + 2 * i], n_i)
            cm = self._draw_class_counts(M[:, 2 + 2 * i], n_i)
            if partial_MSE:
                grad[i] = np.mean((p0 - 1.0) * (self._success(cp, y_eval) / n_i - self._success(cm, y_eval) / n_i))
            else:
                grad[i] = np.mean(np.sum(err * (cp / n_i[:, None] - cm / n_i[:, None]), axis=1))
        return grad

    def _device_class_sums(self, params, n_shots_list, wt_dist, eval_ind, y_eval, kind, want_var=False, want_gd=False):
        """The per-component batch sums of a classifier oracle with everything behind the marginals on the
        GPU: one multinomial per (sample, shift) from the library's device sampler
        (myQNNs_classifier.py:607-622 for all 2P+1 instances of every sample at once), outcome 3 re-dealt
        by a second multinomial when class_unbias, then one kernel (vqa_class_stats_dev) that forms the
        class frequencies, the per-sample gradient terms and variance estimates and sums them over the
        batch.  Same estimators, counters and shot totals as the host path; the random stream is the
        sampler's, seeded from NumPy's.  With a process group the batch is split into blocks over the
        ranks (rows of the sampler numbered in the whole batch, so the counts do not depend on the split)
        and the sums are added with one all-reduce.  Returns (sums [P, 4] numpy, per-sample gradient terms
        of this rank's rows [P, B_local] device tensor or None, B), or None when the circuit does not
        compute on a GPU."""
        if not hasattr(self.circuit, "handle"):
            return None
        import torch

        from .sampling import sample_counts
        from .stats import class_stats

        P = len(params)
        Bg, K = len(eval_ind), 2 * P + 1
        sh = self._shard(Bg)
        rows = slice(None) if sh is None else slice(sh.lo, sh.hi)
        row0 = 0 if sh is None else sh.lo
        theta = np.asarray(params[:self.n_c_params], dtype=np.float64)
        n_all = np.floor(np.asarray(n_shots_list, dtype=float)[:P, None] * wt_dist[None, :])    # [P, B]
        n0 = np.floor(np.max(n_shots_list) * wt_dist)
        seed = int(np.random.randint(0, 2 ** 31 - 1))
        if sh is not None:
            seed = int(sh.bcast_ints([seed])[0])
        _, M = self.circuit.psr_full(theta, self._handle("train"), sel=np.asarray(eval_ind[rows], np.int32), obs_list=[],
                                     n_marg_bits=2)
        if not hasattr(M, "detach"):
            return None
        B = int(M.shape[0])
        dev = M.device
        n_inst = torch.empty((B, K), dtype=torch.int64, device=dev)
        n_inst[:, 0] = torch.from_numpy(n0[rows].astype(np.int64)).to(dev)
        nt = torch.from_numpy(n_all[:, rows].astype(np.int64).T.copy()).to(dev)
        n_inst[:, 1::2] = nt
        n_inst[:, 2::2] = nt
        k = sample_counts(M.reshape(B * K, 4).clamp(0.0, 1.0), n_inst.reshape(-1), seed, row_offset=row0 * K)
        if self.class_unbias:
            third = torch.full((B * K, 3), 1.0 / 3.0, dtype=torch.float64, device=dev)
            c = k[:, :3] + sample_counts(third, k[:, 3], seed, stream_id=1, row_offset=row0 * K)
        else:
            c = torch.stack([k[:, 0] + k[:, 3], k[:, 1], k[:, 2]], dim=1)
        sums, gd = class_stats(c.reshape(B, K, 3), n_inst, np.asarray(y_eval, dtype=np.int32)[rows], P, kind,
                               want_variances=want_var, want_grad_data=want_gd)
        S = sums.cpu().numpy() if sh is None else sh.reduce(sums)[0]
        self.circuit_counter[0] += Bg * K
        self.call_counter[0] += Bg * K
        self.shots_counter[0] += int(n0.sum() + 2 * n_all.sum())
        return S, gd, Bg

    def _grad_on_device(self, params, n_shots_list, wt_dist, eval_ind, y_eval, y_1hot, partial_MSE):
        """MSE_grad_eval on the GPU (see _device_class_sums); None when the circuit does not compute on one."""
        from .stats import FULL_MSE, PARTIAL_MSE

        got = self._device_class_sums(params, n_shots_list, wt_dist, eval_ind, y_eval, PARTIAL_MSE if partial_MSE else FULL_MSE)
        if got is None:
            return None
        S, _, B = got
        out = np.zeros_like(params)
        out[:] = S[:, 0] / B
        return out

    @staticmethod
    def _ev_from_sums(S, B):
        """Batch mean of the conditional-variance estimate, or of its fallback form where that mean is negative
        (myQNNs_classifier.py:875-885, 1166-1168)."""
        ev = S[:, 1] / B
        return np.where(ev < 0, S[:, 2] / B, ev)

    def _estimate_MSEgrad_mean_condVar(self, pp, pp2, varpp, pm, pm2, varpm, p0, p02, varp0, y_eval_1hot=None,
                                       second_order=False, partial_MSE=True):
        """myQNNs_classifier.py:861-886."""
        vardp = varpp + varpm
        if partial_MSE:
            EV = np.mean((p02 - 2 * p0 + 1.0) * vardp + varp0 * (pp2 + pm2 - 2.0 * pp * pm))
            if EV < 0:
                EV = np.mean((p0 - 1.0) ** 2 * vardp + varp0 * (pp - pm) ** 2)
            return EV
        term1 = (p02 + (-2 * p0 + 1) * y_eval_1hot) * vardp
        term2 = varp0 * (pp2 + pm2 - 2.0 * pp * pm)
        EV = np.mean(np.sum(term1 + term2, axis=1))
        if EV < 0:
            EV = np.mean(np.sum((p0 - y_eval_1hot) ** 2 * vardp + varp0 * (pp - pm) ** 2, axis=1))
        return EV

    def MSE_iEval(self, params, n_shots_list, option_min_approx=None, option_weight=None, mini_batch_size=None,
                  Max_n_shots0=True, second_order=True, partial_MSE=False, use_QEM=False, **unused):
        """myQNNs_classifier.py:888-999 without error mitigation."""
        if use_QEM:
            raise NotImplementedError("error mitigation is outside this backend")
        self.partial_MSE = partial_MSE
        grad = np.zeros_like(params)
        gvar_list_EV = np.zeros_like(params)
        wt_dist, eval_ind, y_eval, y_1hot = self._setup(mini_batch_size, option_weight)
        if getattr(self, "device_statistics", False):
            from .stats import FULL_MSE, PARTIAL_MSE

            got = self._device_class_sums(params, n_shots_list, wt_dist, eval_ind, y_eval, PARTIAL_MSE if partial_MSE else FULL_MSE,
                                          want_var=True, want_gd=True)
            if got is not None:
                S, gd, B = got
                grad[:] = S[:, 0] / B
                gvar_list_EV[:] = self._ev_from_sums(S, B)
                z = np.zeros_like(params)
                # (with a process group 'grad_data_list' holds this rank's rows of the batch)
                return {"grad": grad, "gvar_list_EV": gvar_list_EV, "gvar_list_EVi": z.copy(), "gvar_list_EV0": z.copy(),
                        "gvar_list_b": z.copy(), "grad_data_list": gd}
        grad_data_list = np.zeros((len(params), len(eval_ind)))
        M = self._psr_marginals(params, eval_ind)
        n0 = np.floor(np.max(n_shots_list) * wt_dist).astype(int)
        c0 = self._draw_class_counts(M[:, 0], n0)
        if partial_MSE:
            p0, p02, varp0 = self._count_stats(self._success(c0, y_eval), n0)
        else:
            p0, p02, varp0 = self._count_stats(c0, n0)
            err = p0 - y_1hot
        for i in range(len(params)):
            n_i = np.floor(n_shots_list[i] * wt_dist).astype(int)
            cp = self._draw_class_counts(M[:, 1 + 2 * i], n_i)
            cm = self._draw_class_counts(M[:, 2 + 2 * i], n_i)
            if partial_MSE:
                pp, pp2, varpp = self._count_stats(self._success(cp, y_eval), n_i)
                pm, pm2, varpm = self._count_stats(self._success(cm, y_eval), n_i)
                grad_data_list[i] = (p0 - 1.0) * (pp - pm)
                gvar_list_EV[i] = self._estimate_MSEgrad_mean_condVar(pp, pp2, varpp, pm, pm2, varpm, p0, p02, varp0,
                                                                      partial_MSE=True)
            else:
                pp, pp2, varpp = self._count_stats(cp, n_i)
                pm, pm2, varpm = self._count_stats(cm, n_i)
                grad_data_list[i] = np.sum(err * (pp - pm), axis=1)
                gvar_list_EV[i] = self._estimate_MSEgrad_mean_condVar(pp, pp2, varpp, pm, pm2, varpm, p0, p02, varp0,
                                                                      y_eval_1hot=y_1hot, partial_MSE=False)
            grad[i] = np.mean(grad_data_list[i])
        z = np.zeros_like(params)
        return {"grad": grad, "gvar_list_EV": gvar_list_EV, "gvar_list_EVi": z.copy(), "gvar_list_EV0": z.copy(),
                "gvar_list_b": z.copy(), "grad_data_list": grad_data_list}

    def _success_outcomes(self, p_success, n_s, marg=None, y_eval=None):
        """0/1 outcome arrays of 'the true class was read' (myQNNs_classifier.py:818-822)."""
        self.circuit_counter[0] += len(n_s)
        self.call_counter[0] += len(n_s)
        self.shots_counter[0] += int(np.sum(n_s))
        if self.shot_draws == "per_shot":
            out = [(self._class_outcomes(m, n) == int(c)).astype(float) for m, n, c in zip(marg, n_s, y_eval)]
        else:
            out = [(np.random.random(int(n)) < p).astype(float) for p, n in zip(p_success, n_s)]
        return out, np.array([o.sum() for o in out])

    def MSE_iEval_iRandom(self, params, n_shots_list, option_weight=None, partial_MSE=True):
        """myQNNs_classifier.py:1017-1053."""
        grad = np.zeros_like(params)
        gvar_list = np.zeros_like(params)
        if option_weight is None:
            wt_dist = self._uniform_wt_dist.copy()
        else:
            wt_dist = np.array(option_weight) / np.sum(option_weight)
        M_tot = self.num_train_data
        theta = np.asarray(params[:self.n_c_params], dtype=np.float64)
        if getattr(self, "device_statistics", False) and self.shot_draws != "per_shot" and hasattr(self.circuit, "handle"):
            return self._irandom_on_device(params, n_shots_list, wt_dist, theta, kind=1)
        for i in range(len(params)):
            n_shots_data_list = np.random.multinomial(n_shots_list[i], wt_dist)
            sel = np.nonzero(n_shots_data_list != 0)[0].astype(np.int32)
            y_eval = self.y_train[sel]
            n_eval = n_shots_data_list[sel]
            n = n_shots_list[i]
            E = len(sel)
            si = np.tile(sel, 3)
            sp = np.concatenate([np.full(E, -1), np.full(E, i), np.full(E, i)]).astype(np.int32)
            sv = np.concatenate([np.zeros(E), np.full(E, 0.5 * np.pi), np.full(E, -0.5 * np.pi)])
            per_shot = self.shot_draws == "per_shot"
            e, m = self.circuit.eval_batch(theta, self._handle("train"), sample_idx=si, shift_param=sp, shift_val=sv,
                                           obs_list=self.class_observables(), n_marg_bits=2 if per_shot else 0)
            p = np.clip(_np(e)[np.arange(3 * E), np.tile(np.asarray(y_eval, dtype=int), 3)], 0.0, 1.0)
            m = np.clip(_np(m), 0.0, 1.0) if per_shot else [None] * (3 * E)
            out0, sum0 = self._success_outcomes(p[:E], n_eval, m[:E], y_eval)
            outp, sump = self._success_outcomes(p[E:2 * E], n_eval, m[E:2 * E], y_eval)
            outm, summ = self._success_outcomes(p[2 * E:], n_eval, m[2 * E:], y_eval)
            grad[i] = np.sum((sum0 / (1.0 + (n - 1) / M_tot) - 1.0) * (sump - summ) / n)
            g = (np.concatenate(out0) - 1.0) * (np.concatenate(outp) - np.concatenate(outm))
            gvar_list[i] = np.var(g, ddof=1)
        return {"grad": grad, "gvar_list": gvar_list}

    # ---- cross entropy (myQNNs_classifier.py:1158-1269) ---------------------------------------
    def _estimate_CEgrad_mean_condVar(self, theta, pp, pp2, varpp, pm, pm2, varpm, second_order=False):
        term1 = theta ** 2 * (varpp + varpm)
        var_theta = theta ** 2 * (theta - 1.0)
        EV = 0.25 * np.mean(term1 + var_theta * (pp2 + pm2 - 2.0 * pp * pm))
        if EV < 0:
            EV = 0.25 * np.mean(term1 + var_theta * (pp - pm) ** 2)
        return EV

    def _CE_sweep(self, params, n_shots_list, mini_batch_size, option_weight, L2_reg_lmd, with_var):
        grad = np.zeros_like(params)
        gvar_list_EV = np.zeros_like(params)
        wt_dist, eval_ind, y_eval, _ = self._setup(mini_batch_size, option_weight)
        if getattr(self, "device_statistics", False):
            from .stats import CROSS_ENTROPY

            got = self._device_class_sums(params, n_shots_list, wt_dist, eval_ind, y_eval, CROSS_ENTROPY, want_var=with_var,
                                          want_gd=with_var)
            if got is not None:
                S, gd, B = got
                grad[:] = S[:, 0] / B + L2_reg_lmd * np.asarray(params, dtype=float)
                if with_var:
                    gvar_list_EV[:] = self._ev_from_sums(S, B)
                # as in the reference, grad_data_list holds the last component's per-sample values
                return grad, gvar_list_EV, (gd[-1] if gd is not None else None)
        M = self._psr_marginals(params, eval_ind)
        n0 = np.floor(np.max(n_shots_list) * wt_dist).astype(int)
        count0 = self._success(self._draw_class_counts(M[:, 0], n0), y_eval)
        theta = (n0 + 1) / (count0 + 1)
        grad_data_list = None
        for i in range(len(params)):
            n_i = np.floor(n_shots_list[i] * wt_dist).astype(int)
            countp = self._success(self._draw_class_counts(M[:, 1 + 2 * i], n_i), y_eval)
            countm = self._success(self._draw_class_counts(M[:, 2 + 2 * i], n_i), y_eval)
            grad_data_list = -0.5 * (countp - countm) / (count0 + 1) * ((n0 + 1) / (n_i + 1))
            grad[i] = np.mean(grad_data_list) + L2_reg_lmd * params[i]
            if with_var:
                pp, pp2, varpp = self._count_stats(countp, n_i)
                pm, pm2, varpm = self._count_stats(countm, n_i)
                gvar_list_EV[i] = self._estimate_CEgrad_mean_condVar(theta, pp, pp2, varpp, pm, pm2, varpm)
        return grad, gvar_list_EV, grad_data_list

    def CE_grad_eval(self, params, n_shots_list, mini_batch_size=None, option_min_approx=None, option_weight=None,
                     L2_reg_lmd=0.4, second_order=False, Max_n_shots0=True):
        return self._CE_sweep(params, n_shots_list, mini_batch_size, option_weight, L2_reg_lmd, False)[0]

    def CE_iEval(self, params, n_shots_list, mini_batch_size=None, option_min_approx=None, option_weight=None,
                 L2_reg_lmd=0.01, second_order=False, Max_n_shots0=True):
        grad, ev, gdl = self._CE_sweep(params, n_shots_list, mini_batch_size, option_weight, L2_reg_lmd, True)
        z = np.zeros_like(params)
        # as in the reference, grad_data_list holds the last component's per-sample values
        return {"grad": grad, "gvar_list_EV": ev, "gvar_list_EVi": z.copy(), "gvar_list_EV0": z.copy(),
                "gvar_list_b": z.copy(), "grad_data_list": gdl}

    def CE_iEval_iRandom(self):
        """Empty in the reference (myQNNs_classifier.py:1272-1273)."""

    def CE_loss_eval(self):
        """Empty in the reference (myQNNs_classifier.py:1275-1276)."""

    # MSE_loss_eval: inherited from the regressor as in the reference (myQNNs_base.py:1397-1419 through
    # myQNNs_classifier.py:48): a Z-readout estimator scaled by params[-1], which is a CIRCUIT parameter
    # here -- not a loss of the class probabilities; kept so that the optimizers' 'loss' slot behaves
    # as it does there.
