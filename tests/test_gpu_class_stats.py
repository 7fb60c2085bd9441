"""The classifier's estimator epilogue on the GPU (vqa_class_stats_dev) against the host algebra of
vqa_amd.qnn_classifier (the reference's formulas, myQNNs_classifier.py:623-627, 861-886, 1158-1169) on the SAME class
counts, then the oracles that use it against the host path: same counters, same statistics."""
import numpy as np
import pytest

from test_gpu_host_layer import pair
from vqa_amd.qml_utils import create_ibm_HEA_ansatz
from vqa_amd.qnn_classifier import myQNN_classifier

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_class_stats_kernel_equals_host_algebra(kind):
    import torch
    from vqa_amd.stats import class_stats

    (q, _), _ = pair(create_ibm_HEA_ansatz, 5, 1, myQNN_classifier, n_train=8)
    rng = np.random.default_rng(kind)
    B, P = 37, 9
    K = 2 * P + 1
    shots = rng.integers(4, 9, (B, K))                       # few shots: negative estimates occur
    shots[:, 2::2] = shots[:, 1::2]
    counts = np.zeros((B, K, 3), np.int64)
    for b in range(B):
        for k in range(K):
            counts[b, k] = rng.multinomial(shots[b, k], rng.dirichlet(np.ones(3)))
    y = rng.integers(0, 3, B)
    onehot = np.eye(3)[y]
    S, gd = class_stats(torch.from_numpy(counts).cuda(), torch.from_numpy(shots).cuda(), y, P, kind, want_variances=True,
                        want_grad_data=True)
    S, gd = S.cpu().numpy(), gd.cpu().numpy()
    n0 = shots[:, 0]
    for i in range(P):
        n_i = shots[:, 1 + 2 * i]
        cp, cm, c0 = counts[:, 1 + 2 * i].astype(float), counts[:, 2 + 2 * i].astype(float), counts[:, 0].astype(float)
        if kind == 2:
            s0, sp, sm = q._success(c0, y), q._success(cp, y), q._success(cm, y)
            want_gd = -0.5 * (sp - sm) / (s0 + 1) * ((n0 + 1) / (n_i + 1))
            theta = (n0 + 1) / (s0 + 1)
            pp, pp2, vpp = q._count_stats(sp, n_i)
            pm, pm2, vpm = q._count_stats(sm, n_i)
            term1 = theta ** 2 * (vpp + vpm)
            vt = theta ** 2 * (theta - 1.0)
            main, alt = 0.25 * (term1 + vt * (pp2 + pm2 - 2 * pp * pm)), 0.25 * (term1 + vt * (pp - pm) ** 2)
        elif kind == 0:
            p0, p02, v0 = q._count_stats(q._success(c0, y), n0)
            pp, pp2, vpp = q._count_stats(q._success(cp, y), n_i)
            pm, pm2, vpm = q._count_stats(q._success(cm, y), n_i)
            want_gd = (p0 - 1.0) * (pp - pm)
            main = (p02 - 2 * p0 + 1.0) * (vpp + vpm) + v0 * (pp2 + pm2 - 2.0 * pp * pm)
            alt = (p0 - 1.0) ** 2 * (vpp + vpm) + v0 * (pp - pm) ** 2
        else:
            p0, p02, v0 = q._count_stats(c0, n0)
            pp, pp2, vpp = q._count_stats(cp, n_i)
            pm, pm2, vpm = q._count_stats(cm, n_i)
            want_gd = np.sum((p0 - onehot) * (pp - pm), axis=1)
            main = np.sum((p02 + (-2 * p0 + 1) * onehot) * (vpp + vpm) + v0 * (pp2 + pm2 - 2.0 * pp * pm), axis=1)
            alt = np.sum((p0 - onehot) ** 2 * (vpp + vpm) + v0 * (pp - pm) ** 2, axis=1)
        assert np.max(np.abs(gd[i] - want_gd)) < 1e-12
        for col, want in ((0, want_gd), (1, main), (2, alt)):
            assert abs(S[i, col] - want.sum()) < 1e-10 * max(1.0, np.abs(want).sum()), (i, col)
        # the finished estimate is what the host's estimator returns on these moments
        ev = S[i, 1] / B if S[i, 1] >= 0 else S[i, 2] / B
        if kind == 2:
            host = q._estimate_CEgrad_mean_condVar(theta, pp, pp2, vpp, pm, pm2, vpm)
        else:
            host = q._estimate_MSEgrad_mean_condVar(pp, pp2, vpp, pm, pm2, vpm, p0, p02, v0, y_eval_1hot=onehot,
                                                    partial_MSE=kind == 0)
        assert abs(ev - host) < 1e-10 * max(1.0, abs(host))


@pytest.mark.parametrize("fn,kw", [("MSE_iEval", dict(partial_MSE=True)), ("MSE_iEval", dict(partial_MSE=False)),
                                   ("CE_iEval", dict()), ("CE_grad_eval", dict())])
def test_classifier_oracles_on_device_match_host_statistics(fn, kw):
    """MSE_iEval / CE_grad_eval / CE_iEval with device_statistics: the host path's counters and shot totals;
    gradient and variance estimates with the host path's mean (and, for the gradient, scatter)."""
    (g, p), (o, _) = pair(create_ibm_HEA_ansatz, 5, 1, myQNN_classifier, n_train=24, class_unbias=True)
    g.device_statistics = True
    shots = np.full(g.n_params, 300)
    N = len(g.x_train)
    call = lambda q: getattr(q, fn)(p, shots, mini_batch_size=N, **kw)
    np.random.seed(5)
    a = call(g)
    np.random.seed(5)
    b = call(o)
    assert (g.shots_counter[0], g.call_counter[0], g.circuit_counter[0]) == \
           (o.shots_counter[0], o.call_counter[0], o.circuit_counter[0])
    if isinstance(a, dict):
        assert set(a) == set(b)
        assert hasattr(a["grad_data_list"], "is_cuda")          # the per-sample terms stay on the GPU
    R = 80
    gd, gh = [call(g) for _ in range(R)], [call(o) for _ in range(R)]
    keys = [("grad", True), ("gvar_list_EV", False)] if isinstance(a, dict) else [(None, True)]
    for key, scatter in keys:
        d = np.array([r[key] if key else r for r in gd])
        h = np.array([r[key] if key else r for r in gh])
        sd = h.std(axis=0) + 1e-12
        assert np.max(np.abs(d.mean(axis=0) - h.mean(axis=0)) / sd) < 6 * np.sqrt(2.0 / R), key
        if scatter:
            assert np.all(d.std(axis=0) < 1.8 * sd) and np.all(d.std(axis=0) > sd / 1.8)
