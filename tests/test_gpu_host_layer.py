"""Host layer on the HIP circuit against the same layer on the CPU-oracle
circuit (same NumPy seed), and against the golden trajectories."""
import numpy as np
import pytest

from oracle_circuit import OracleCircuit
from test_host_layer import run_golden_case
from vqa_amd import LearningCircuit
from vqa_amd.qml_utils import create_ibm_HEA_ansatz, create_jerbi_ansatz
from vqa_amd.qnn_base import myQNN_base
from vqa_amd.qnn_classifier import myQNN_classifier

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["Adam", "SGD", "SGDM"])
def test_golden_trajectory_hip_circuit(name):
    # parameters within 1e-8 of the reference optimizer's trajectory, shot totals exact
    run_golden_case(name, LearningCircuit, 1e-8)


def pair(builder, n, depth, cls, n_train=20, seed=1, **kw):
    out = []
    for circuit_cls in (LearningCircuit, OracleCircuit):
        rng = np.random.default_rng(seed)
        c = builder(n, depth, circuit_cls=circuit_cls)
        x = rng.uniform(0, 1, (n_train + 8, n))
        if cls is myQNN_classifier:
            y = rng.integers(0, 3, n_train + 8)
            q = cls(n, c, x[:n_train], y[:n_train], x[n_train:], y[n_train:], **kw)
            p = rng.uniform(-3, 3, q.n_params)
        else:
            q = cls(n, c, min_approx_shots=0, **kw)
            tp = np.concatenate([rng.uniform(-3, 3, q.n_c_params), [1.2]])
            y = np.sin(x.sum(axis=1)) + 0 * tp[0]
            q.set_data(x[:n_train], y[:n_train], x[n_train:], y[n_train:])
            p = np.concatenate([rng.uniform(-3, 3, q.n_c_params), [0.8]])
        out.append((q, p))
    return out


@pytest.mark.parametrize("builder,n,depth", [(create_ibm_HEA_ansatz, 6, 2), (create_jerbi_ansatz, 5, 2),
                                             (create_ibm_HEA_ansatz, 10, 1)])
def test_regressor_oracles_match(builder, n, depth):
    (g, p), (o, _) = pair(builder, n, depth, myQNN_base)
    shots = np.full(g.n_params, 300)
    assert np.max(np.abs(g.exact_predict(g.x_test, p) - o.exact_predict(o.x_test, p))) < 1e-10
    a, b = g.exact_MSE(p, ret_both=True), o.exact_MSE(p, ret_both=True)
    assert abs(a["test"] - b["test"]) < 1e-10 and abs(a["train"] - b["train"]) < 1e-10
    for fn, kw in [("MSE_grad_eval", dict(mini_batch_size=7)), ("MSE_loss_eval", dict(mini_batch_size=7)),
                   ("MSE_iEval", dict(mini_batch_size=7))]:
        res = []
        for q in (g, o):
            np.random.seed(77)
            q.remain_ind = []
            q.shots_counter[0] = q.call_counter[0] = q.circuit_counter[0] = 0
            arg = shots if fn != "MSE_loss_eval" else 300
            res.append((getattr(q, fn)(p, arg, **kw), q.shots_counter[0], q.call_counter[0], q.circuit_counter[0]))
        assert res[0][1:] == res[1][1:]
        if isinstance(res[0][0], dict):
            for k in res[0][0]:
                assert np.max(np.abs(res[0][0][k] - res[1][0][k])) < 1e-8, k
        else:
            assert np.max(np.abs(np.asarray(res[0][0]) - np.asarray(res[1][0]))) < 1e-8
    # exact mode: no random numbers at all
    g.remain_ind, o.remain_ind = [], []
    ge = g.MSE_grad_eval(p, shots, mini_batch_size=len(g.x_train), option_min_approx=-10)
    oe = o.MSE_grad_eval(p, shots, mini_batch_size=len(o.x_train), option_min_approx=-10)
    assert np.max(np.abs(ge - oe)) < 1e-10  # full-set means do not depend on the permutation


@pytest.mark.parametrize("unbias", [True, False])
def test_classifier_oracles_match(unbias):
    (g, p), (o, _) = pair(create_ibm_HEA_ansatz, 6, 1, myQNN_classifier, class_unbias=unbias)
    assert np.max(np.abs(g.exact_prob_dist_data_by_data(g.x_test, p) - o.exact_prob_dist_data_by_data(o.x_test, p))) < 1e-10
    a, b = g.exact_predict_error_and_MSE(p), o.exact_predict_error_and_MSE(p)
    assert all(abs(a[k] - b[k]) < 1e-10 for k in a)
    idx = np.arange(len(g.x_train))
    assert np.max(np.abs(g._psr_marginals(p, idx) - o._psr_marginals(p, idx))) < 1e-10
    shots = np.full(g.n_params, 200)
    for fn in ("MSE_grad_eval", "CE_grad_eval"):
        res = []
        for q in (g, o):
            np.random.seed(5)
            q.remain_ind = []
            res.append(getattr(q, fn)(p, shots, mini_batch_size=6))
        # same seed, probabilities equal to 1e-10: the drawn counts coincide
        assert np.max(np.abs(res[0] - res[1])) < 1e-8


def test_fit_runs_on_gpu_and_pickles(tmp_path):
    (g, p), _ = pair(create_ibm_HEA_ansatz, 4, 1, myQNN_base)
    np.random.seed(0)
    g.compile(optimizer="Adam", hpara=dict(LR_rule=0.05, s_list_rule=np.full(g.n_params, 100), batch_size_rule=6), init_params=p)
    res = g.fit(options=dict(max_iter=15), auto_save=True, save_directory=str(tmp_path))
    assert res["func_evol"][-1]["train"] < res["func_evol"][0]["train"]
    assert len(list(tmp_path.iterdir())) == 1


def test_device_statistics_clt_gradient():
    """CLT-mode gradient with the deviates drawn on the GPU: same counters as the host path,
    and the same values within the shot noise (large shot numbers)."""
    (g, p), (o, _) = pair(create_ibm_HEA_ansatz, 6, 2, myQNN_base, n_train=64)
    g.device_statistics = True
    shots = np.full(g.n_params, 2_000_000)
    np.random.seed(3)
    gd = g.MSE_grad_eval(p, shots, mini_batch_size=64)
    np.random.seed(3)
    gh = o.MSE_grad_eval(p, shots, mini_batch_size=64)
    assert (g.shots_counter[0], g.call_counter[0], g.circuit_counter[0]) == \
           (o.shots_counter[0], o.call_counter[0], o.circuit_counter[0])
    exact = o.MSE_grad_eval(p, shots, mini_batch_size=64, option_min_approx=-10)
    # both are the exact gradient of their (different) mini-batches plus noise of order 1e-3 / sqrt(64)
    assert np.all(np.isfinite(gd)) and np.max(np.abs(gd)) < 10
    g.remain_ind, o.remain_ind = [], []
    np.random.seed(4)
    gd2 = g.MSE_grad_eval(p, shots, mini_batch_size=len(g.x_train))
    np.random.seed(4)
    ex2 = o.MSE_grad_eval(p, shots, mini_batch_size=len(o.x_train), option_min_approx=-10)
    assert np.max(np.abs(gd2 - ex2)) < 5e-3


def test_device_statistics_sampled_gradient():
    """Finite-shot gradient with the outcome counts drawn on the GPU (device sampler): the host
    path's counters and shot totals, an unbiased estimate of the exact gradient (mean of repeated
    draws), and its scatter as the host path's."""
    (g, p), (o, _) = pair(create_ibm_HEA_ansatz, 6, 2, myQNN_base, n_train=32)
    g.device_statistics = True
    g.min_approx_shots = o.min_approx_shots = -1
    shots = np.full(g.n_params, 400)
    N = len(g.x_train)
    np.random.seed(3)
    g.MSE_grad_eval(p, shots, mini_batch_size=N)
    np.random.seed(3)
    o.MSE_grad_eval(p, shots, mini_batch_size=N)
    assert (g.shots_counter[0], g.call_counter[0], g.circuit_counter[0]) == \
           (o.shots_counter[0], o.call_counter[0], o.circuit_counter[0])
    exact = o.MSE_grad_eval(p, shots, mini_batch_size=N, option_min_approx=-10)
    R = 60
    gd = np.array([g.MSE_grad_eval(p, shots, mini_batch_size=N) for _ in range(R)])
    gh = np.array([o.MSE_grad_eval(p, shots, mini_batch_size=N) for _ in range(R)])
    sd = gh.std(axis=0) + 1e-12
    assert np.max(np.abs(gd.mean(axis=0) - exact) / sd) < 6 / np.sqrt(R) * 1.5
    assert np.all(gd.std(axis=0) < 1.8 * sd) and np.all(gd.std(axis=0) > sd / 1.8)


@pytest.mark.parametrize("unbias,partial", [(True, True), (False, False), (True, False)])
def test_classifier_device_statistics_gradient(unbias, partial):
    """Classifier MSE gradient with the class counts drawn by the device sampler: the host path's
    counters and shot totals, an unbiased estimate of the infinite-shot gradient (mean of repeated
    draws against the mean of the host path's draws), and the host path's scatter."""
    (g, p), (o, _) = pair(create_ibm_HEA_ansatz, 5, 1, myQNN_classifier, n_train=24, class_unbias=unbias)
    g.device_statistics = True
    shots = np.full(g.n_params, 300)
    N = len(g.x_train)
    np.random.seed(5)
    g.MSE_grad_eval(p, shots, mini_batch_size=N, partial_MSE=partial)
    np.random.seed(5)
    o.MSE_grad_eval(p, shots, mini_batch_size=N, partial_MSE=partial)
    assert (g.shots_counter[0], g.call_counter[0], g.circuit_counter[0]) == \
           (o.shots_counter[0], o.call_counter[0], o.circuit_counter[0])
    R = 80
    gd = np.array([g.MSE_grad_eval(p, shots, mini_batch_size=N, partial_MSE=partial) for _ in range(R)])
    gh = np.array([o.MSE_grad_eval(p, shots, mini_batch_size=N, partial_MSE=partial) for _ in range(R)])
    sd = gh.std(axis=0) + 1e-12
    assert np.max(np.abs(gd.mean(axis=0) - gh.mean(axis=0)) / sd) < 6 * np.sqrt(2.0 / R)
    assert np.all(gd.std(axis=0) < 1.8 * sd) and np.all(gd.std(axis=0) > sd / 1.8)


@pytest.mark.parametrize("cls", [myQNN_base, myQNN_classifier])
def test_device_irandom_statistics(cls):
    """MSE_iEval_iRandom with the outcome counts of every (component, sample) row drawn and reduced
    by the device kernel: the host path's counters and shot totals, gradient estimates with the host
    path's mean and scatter, variance estimates with the host path's mean."""
    kw = dict(class_unbias=True) if cls is myQNN_classifier else {}
    (g, p), (o, _) = pair(create_ibm_HEA_ansatz, 5, 1, cls, n_train=24, **kw)
    g.device_statistics = True
    if cls is myQNN_base:
        g.min_approx_shots = o.min_approx_shots = -1
    shots = np.full(g.n_params, 600)
    shots[0] = 900
    np.random.seed(5)
    a = g.MSE_iEval_iRandom(p, shots)
    np.random.seed(5)
    b = o.MSE_iEval_iRandom(p, shots)
    assert set(a) == set(b) and a["grad"].shape == b["grad"].shape
    assert (g.shots_counter[0], g.call_counter[0], g.circuit_counter[0]) == \
           (o.shots_counter[0], o.call_counter[0], o.circuit_counter[0])
    R = 80
    gd = [g.MSE_iEval_iRandom(p, shots) for _ in range(R)]
    gh = [o.MSE_iEval_iRandom(p, shots) for _ in range(R)]
    for key, scatter in (("grad", True), ("gvar_list", False)):
        d = np.array([r[key] for r in gd])
        h = np.array([r[key] for r in gh])
        sd = h.std(axis=0) + 1e-12
        assert np.max(np.abs(d.mean(axis=0) - h.mean(axis=0)) / sd) < 6 * np.sqrt(2.0 / R), key
        if scatter:
            assert np.all(d.std(axis=0) < 1.8 * sd) and np.all(d.std(axis=0) > sd / 1.8)
    # few shots: most samples receive none, rows are ragged per component
    few = np.full(g.n_params, 7)
    np.random.seed(6)
    c = g.MSE_iEval_iRandom(p, few)
    assert np.all(np.isfinite(c["grad"])) and np.all(np.isfinite(c["gvar_list"])) and np.all(c["gvar_list"] >= 0)


@pytest.mark.parametrize("second_order,max_n0", [(True, True), (False, False)])
def test_device_ieval_equals_host_algebra(second_order, max_n0):
    """MSE_iEval evaluated on the GPU (all parameters at once, fallbacks included) against the
    host's per-parameter NumPy algebra fed the SAME per-instance estimates; small shot numbers so
    that some means come out negative and the reference's fallback branches are taken."""
    import torch

    (g, p), _ = pair(create_ibm_HEA_ansatz, 5, 1, myQNN_base, n_train=24)
    (h, _), _ = pair(create_ibm_HEA_ansatz, 5, 1, myQNN_base, n_train=24)
    N = len(g.x_train)
    g.remain_ind, h.remain_ind = np.arange(N), np.arange(N)
    rng = np.random.default_rng(5)
    shots = rng.integers(5, 12, g.n_params)
    E = h._psr_all(p, np.arange(N))                                            # [N, 2P+1] exact values
    M = np.clip(E + rng.normal(size=E.shape) * 0.4, -1, 1)                     # any estimates will do
    cols = iter(range(E.shape[1]))
    orig = h._draw_means
    h._draw_means = lambda exact, n_s, ma, marg=None: (orig(exact, n_s, ma, marg), M[:, next(cols)])[1]
    want = h.MSE_iEval(p, shots, mini_batch_size=N, option_min_approx=0, Max_n_shots0=max_n0, second_order=second_order)
    ma, wt, idx, y = g._batch_setup(N, 0, None)
    got = g._ieval_on_device(p, shots, wt, idx, y, 0, N, max_n0, second_order, _estimates=torch.from_numpy(M).cuda())
    for k in ("grad", "gvar_list_EV", "gvar_list_EVi", "gvar_list_EV0", "gvar_list_b"):
        assert np.max(np.abs(got[k] - want[k])) < 1e-10 * max(1.0, np.max(np.abs(want[k]))), k
    assert np.max(np.abs(got["grad_data_list"].cpu().numpy() - want["grad_data_list"])) < 1e-12
    assert (g.shots_counter[0], g.call_counter[0], g.circuit_counter[0]) == \
           (h.shots_counter[0], h.call_counter[0], h.circuit_counter[0])
    # through the public method: device_statistics routes MSE_iEval to the device
    g.device_statistics = True
    g.remain_ind = np.arange(N)
    np.random.seed(1)
    out = g.MSE_iEval(p, np.full(g.n_params, 5000), mini_batch_size=N, option_min_approx=0)
    assert hasattr(out["grad_data_list"], "is_cuda") and np.all(np.isfinite(out["gvar_list_EV"]))
    exact = h.MSE_grad_eval(p, shots, mini_batch_size=N, option_min_approx=-10)
    assert np.max(np.abs(out["grad"] - exact)) < 0.2


def test_pauli_sum_target_matches_oracle_circuit():
    from test_vqe_target import make
    vg, p = make(n=10, d=3, min_approx=-10, cls=LearningCircuit)
    vo, _ = make(n=10, d=3, min_approx=-10, cls=OracleCircuit)
    assert abs(vg.exact_energy(p) - vo.exact_energy(p)) < 1e-9
    a, b = vg.iEval(p, np.full(vg.n_params, 10)), vo.iEval(p, np.full(vo.n_params, 10))
    assert np.max(np.abs(a["grad"] - b["grad"])) < 1e-9
    assert np.max(np.abs(a["gvar_list"] - b["gvar_list"])) < 1e-8


def test_iris_classifier_trains_on_gpu():
    """The 4-qubit Iris configuration end to end: RY encoding, 3 layers of RY/RZ + CZ ring,
    partial-MSE gradient with finite shots, built-in Adam; the exact error rate must not get
    worse and the exact loss must go down."""
    from sklearn.datasets import load_iris
    from vqa_amd.qml_utils import create_ry_rz_cz_ansatz

    data = load_iris()
    x = (data.data - data.data.min(0)) / (data.data.max(0) - data.data.min(0)) * np.pi
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(x))
    tr, te = perm[:105], perm[105:]
    c = create_ry_rz_cz_ansatz(4, 3, seed=0, circuit_cls=LearningCircuit)
    clf = myQNN_classifier(4, c, x[tr], data.target[tr], x[te], data.target[te])
    p0 = rng.uniform(0, 2 * np.pi, clf.n_params)
    np.random.seed(1)
    clf.compile(optimizer="Adam", loss="mean_squared_error", partial_MSE=True,
                hpara=dict(LR_rule=0.1, s_list_rule=np.full(clf.n_params, 200), batch_size_rule=32), init_params=p0)
    res = clf.fit(options=dict(max_iter=40), auto_save=False)
    first, last = res["func_evol"][0], res["func_evol"][-1]
    assert last["train loss"] < first["train loss"] - 0.05
    assert last["train"] <= first["train"] + 1e-9
    assert res["total_shots"][-1] == 40 * 32 * 200 * (2 * clf.n_params + 1)


@pytest.mark.parametrize("partial", [True, False])
def test_class_grad_kernel_matches_numpy(partial):
    """vqa_class_grad_dev against the same frequencies, products and means in NumPy
    (myQNNs_classifier.py:1055-1156 with the counts given)."""
    import ctypes as C

    import torch

    from vqa_amd import _lib

    rng = np.random.default_rng(3)
    B, P = 300, 7
    K = 2 * P + 1
    shots = rng.integers(5, 200, (B, K))
    counts = np.stack([rng.multinomial(n, [0.2, 0.5, 0.3]) for n in shots.reshape(-1)]).reshape(B, K, 3)
    y = rng.integers(0, 3, B)
    q = counts / shots[:, :, None]
    if partial:
        ps = q[np.arange(B), :, y]
        want = ((ps[:, 0] - 1.0)[:, None] * (ps[:, 1::2] - ps[:, 2::2])).mean(0)
    else:
        want = ((q[:, 0, :] - np.eye(3)[y])[:, None, :] * (q[:, 1::2, :] - q[:, 2::2, :])).sum(-1).mean(0)
    dev = torch.device("cuda:0")
    d_c = torch.from_numpy(counts.astype(np.int64)).to(dev)
    d_n = torch.from_numpy(shots.astype(np.int64)).to(dev)
    d_lab = torch.from_numpy(y.astype(np.int32)).to(dev) if partial else None
    d_hot = None if partial else torch.from_numpy(np.eye(3)[y]).to(dev)
    out = torch.empty(P, dtype=torch.float64, device=dev)
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    _lib.check(_lib.lib().vqa_class_grad_dev(0, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), ptr(d_c), ptr(d_n),
                                             B, P, ptr(d_lab), ptr(d_hot), ptr(out)))
    assert np.max(np.abs(out.cpu().numpy() - want)) < 1e-14
    with pytest.raises(_lib.VqaError):
        _lib.check(_lib.lib().vqa_class_grad_dev(0, None, ptr(d_c), ptr(d_n), B, P, None, None, ptr(out)))
