"""TEST INFRASTRUCTURE -- NumPy restatement of the library's estimator epilogue (vqa_mse_stats_shard_dev,
vqa_amd/csrc/kern_stats.hip) for the modes without random numbers: 0 (every estimate is the exact value) and
3 (the caller supplies the estimates).  Same argument list as vqa_amd.stats.mse_stats, so that CPU tests can plug
it into a target object (attribute _stats_fn) and exercise the sharding, the all-reduce and the finishing
algebra of the gradient oracles without a GPU.  The formulas are the reference's: unbiased powers
vqa/QNN/myQNNs_base.py:843-847, gradient terms 1378 / 1393, variance polynomials 1205-1316.
Only tests may import this module."""
import numpy as np

EXACT, GIVEN = 0, 3


def mse_stats(E, y, w, shots=None, weight=None, n_shots0=0.0, mode=EXACT, seed=0, want_variances=False, n0=0.0,
              second_order=False, estimates=None, want_m0=False, want_grad_data=False, row_offset=0, device_sums=False):
    if mode not in (EXACT, GIVEN):
        raise ValueError("the NumPy restatement has no random streams: modes 0 and 3 only")
    E = np.asarray(E, dtype=float)
    B, K = E.shape
    P = (K - 1) // 2
    y = np.asarray(y, dtype=float)
    wt = np.ones(B) if weight is None else np.asarray(weight, dtype=float)
    M = E if mode == EXACT else np.asarray(estimates, dtype=float)
    so = bool(second_order) and n0 != 0.0
    w2 = w * w
    S = np.zeros((P + 1, 8))
    EX = M[:, 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        if mode == EXACT:
            EX_2, EX_3, EX_4, var_0 = EX ** 2, EX ** 3, EX ** 4, 1.0 - EX ** 2
        else:
            n = np.floor(n_shots0 * wt)
            EX_2 = (n / (n - 1.0)) * EX * EX - 1.0 / (n - 1.0)
            EX_3 = (n / (n - 1.0)) * (n / (n - 2.0)) * EX ** 3 - (3.0 * n - 2.0) / (n - 1.0) / (n - 2.0) * EX
            EX_4 = (n / (n - 1.0)) * (n / (n - 2.0)) * (n / (n - 3.0)) * EX ** 4 - \
                ((n / (n - 1.0)) * ((6.0 * n - 8.0) / (n - 2.0)) * EX * EX - 3.0 / (n - 1.0)) / (n - 3.0)
            var_0 = n / (n - 1.0) * (1.0 - EX * EX)
        resid = (w * EX - y) * w
        err2 = y * y - 2.0 * w * y * EX + w2 * EX_2
        alt_err2 = (y - w * EX) ** 2
        var_err = w2 * var_0
        A, Bc = -2.0 * y, 2.0 * w
        gd_rows = np.zeros((P + 1, B))
        gd = 2.0 * (-y * EX + w * EX_2)
        gd_rows[P] = gd
        S[P, 0], S[P, 1] = gd.sum(), (gd * gd).sum()
        if want_variances:
            est = A * A + 4.0 * A * Bc * EX + (4.0 * Bc * Bc - A * A) * EX_2 - 4.0 * A * Bc * EX_3 - 4.0 * Bc * Bc * EX_4
            alt = (A + 2.0 * Bc * EX) ** 2 * var_0
            if so:
                est = est + 2.0 * Bc * Bc * (1.0 - 2.0 * EX + EX_2) / (n0 - 1.0)
                alt = alt + 2.0 * Bc * Bc * var_0 * var_0 / (n0 - 1.0)
            S[P, 2], S[P, 4] = est.sum(), alt.sum()
            S[P, 6] = (A * A * EX_2 + 2.0 * A * Bc * EX_3 + Bc * Bc * EX_4).sum()
        for i in range(P):
            Mp, Mm = M[:, 1 + 2 * i], M[:, 2 + 2 * i]
            d = Mp - Mm
            gd = resid * d
            gd_rows[i] = gd
            S[i, 0], S[i, 1] = gd.sum(), (gd * gd).sum()
            if want_variances:
                n = np.floor(shots[i] * wt)
                ub2 = lambda m: (n / (n - 1.0)) * m * m - 1.0 / (n - 1.0)
                var_p, var_m = n / (n - 1.0) * (1.0 - Mp * Mp), n / (n - 1.0) * (1.0 - Mm * Mm)
                dp2 = ub2(Mp) - 2.0 * Mp * Mm + ub2(Mm)
                var_dpdt = w2 * (var_p + var_m)
                extra = var_err * var_dpdt / n0 if so else 0.0
                S[i, 2] = (err2 * var_dpdt + extra).sum()
                S[i, 3] = (var_err * w2 * dp2).sum()
                S[i, 4] = (alt_err2 * var_dpdt + extra).sum()
                S[i, 5] = (var_err * w2 * d * d).sum()
                S[i, 6] = (err2 * w2 * dp2).sum()
    return S, (EX.copy() if want_m0 else None), (gd_rows if want_grad_data else None)
