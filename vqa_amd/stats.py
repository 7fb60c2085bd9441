"""Estimator epilogue of the regressor's gradient oracles on the GPU: thin wrapper of
vqa_mse_stats_dev (include/vqa_hip.h).  From the matrix of exact expectation values of a
parameter-shift sweep it returns, per gradient component, the batch sums the reference's
MSE_grad_eval / MSE_iEval reduce to (vqa/QNN/myQNNs_base.py:1346-1395, 1421-1596) -- one
hand-written kernel pass instead of the reference's per-sample Python estimators."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

EXACT, CLT, SHOTS, GIVEN = 0, 1, 2, 3
# columns of the returned [P + 1, 8] matrix
GD, GD_SQ, EV_I, EV_0, ALT_I, ALT_0, GD2 = range(7)


def mse_stats(E, y, w, shots=None, weight=None, n_shots0=0.0, mode=EXACT, seed=0, want_variances=False, n0=0.0,
              second_order=False, estimates=None, want_m0=False, want_grad_data=False):
    """E: torch cuda tensor [B, 2P+1] (may be a strided view whose rows are contiguous).  Returns
    (sums [P+1, 8] numpy, m0 [B] tensor or None, grad_data [P+1, B] tensor or None)."""
    import torch

    E = E.contiguous()
    dev = E.device
    B, K = int(E.shape[0]), int(E.shape[1])
    P = (K - 1) // 2
    f64 = lambda a: None if a is None else (a if hasattr(a, "is_cuda") else torch.from_numpy(np.ascontiguousarray(a, np.float64))).to(dev, torch.float64).contiguous()
    y_t, shots_t, weight_t, est_t = f64(y), f64(shots), f64(weight), f64(estimates)
    L = _lib.lib()
    ws = torch.empty(max(int(L.vqa_mse_stats_workspace(B, P)) // 8, 1), dtype=torch.float64, device=dev)
    out = torch.empty((P + 1, 8), dtype=torch.float64, device=dev)
    m0 = torch.empty(B, dtype=torch.float64, device=dev) if want_m0 else None
    gd = torch.empty((P + 1, B), dtype=torch.float64, device=dev) if want_grad_data else None
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(L.vqa_mse_stats_dev(dev.index or 0, C.c_void_p(st), p(E), p(est_t), B, P, p(y_t), float(w), p(shots_t),
                                   p(weight_t), float(n_shots0), int(mode), int(seed), int(bool(want_variances)), float(n0),
                                   int(bool(second_order)), p(ws), p(out), p(m0), p(gd)))
    return out.cpu().numpy(), m0, gd
