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
              second_order=False, estimates=None, want_m0=False, want_grad_data=False, row_offset=0, device_sums=False):
    """E: torch cuda tensor [B, 2P+1] (may be a strided view whose rows are contiguous).  Returns
    (sums [P+1, 8], m0 [B] tensor or None, grad_data [P+1, B] tensor or None); the sums as a numpy array,
    or with device_sums as the device tensor the kernel wrote (no host synchronisation: what a sharded
    caller hands to its all-reduce).  row_offset: number of E's first row in the whole mini-batch when E
    is one rank's shard of it (names the random streams, see vqa_mse_stats_shard_dev)."""
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
    _lib.check(L.vqa_mse_stats_shard_dev(dev.index or 0, C.c_void_p(st), p(E), p(est_t), B, P, p(y_t), float(w), p(shots_t),
                                         p(weight_t), float(n_shots0), int(mode), int(seed), int(bool(want_variances)),
                                         float(n0), int(bool(second_order)), p(ws), p(out), p(m0), p(gd), int(row_offset)))
    return (out if device_sums else out.cpu().numpy()), m0, gd


def irandom_stats(circuit, theta, inputs, obs_list, rows_sample, rows_comp_seg, rows_shots, n_list, w, M_tot, seed,
                  kind=0, has_scale=False, rows_y=None, rows_label=None):
    """Gradient and variance estimates of MSE_iEval_iRandom from ONE explicit instance list on the GPU
    (thin wrapper of vqa_eval_batch_dev + vqa_irandom_stats_dev).  rows_sample[r]: training-set index
    of row r; rows_comp_seg[i] .. [i+1]: the rows of component i (circuit parameter i; with has_scale the
    last component is the scale parameter, whose three instances are all unshifted).  Returns
    out [C, 2] (gradient, variance) as numpy."""
    import torch

    C_ = len(rows_comp_seg) - 1
    R = len(rows_sample)
    seg = np.asarray(rows_comp_seg, np.int32)
    comp = np.repeat(np.arange(C_, dtype=np.int32), np.diff(seg))
    n_circ = C_ - 1 if has_scale else C_
    si = np.repeat(np.asarray(rows_sample, np.int32), 3)
    sp = np.repeat(np.where(comp < n_circ, comp, -1).astype(np.int32), 3)
    sp[0::3] = -1
    sv = np.tile(np.array([0.0, 0.5 * np.pi, -0.5 * np.pi]), R)
    sv[sp < 0] = 0.0
    e, _ = circuit.eval_batch(theta, inputs, sample_idx=si, shift_param=sp, shift_val=sv, obs_list=obs_list, device_out=True)
    dev = e.device
    t = lambda a, dt: None if a is None else torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
    d_y, d_lab = t(rows_y, np.float64), t(rows_label, np.int32)
    d_shots, d_seg, d_n = t(rows_shots, np.int64), t(seg, np.int32), t(n_list, np.float64)
    out = torch.empty((C_, 2), dtype=torch.float64, device=dev)
    p = lambda x: None if x is None else C.c_void_p(x.data_ptr())
    st = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(_lib.lib().vqa_irandom_stats_dev(dev.index or 0, C.c_void_p(st), p(e), int(e.shape[1]), p(d_lab), p(d_y),
                                                p(d_shots), p(d_seg), p(d_n), C_, int(kind), int(bool(has_scale)),
                                                float(w), float(M_tot), int(seed), p(out)))
    return out.cpu().numpy()


PARTIAL_MSE, FULL_MSE, CROSS_ENTROPY = 0, 1, 2


def class_stats(counts, shots, labels, n_params, kind, want_variances=False, want_grad_data=False):
    """The classifier's per-component batch sums from drawn class counts (thin wrapper of
    vqa_class_stats_dev): counts [B, 2P+1, 3] and shots [B, 2P+1] int64 cuda tensors, labels [B].
    Returns (sums [P, 4] device tensor: gradient term, conditional-variance estimate, its fallback
    form; per-sample gradient terms [P, B] device tensor or None)."""
    import torch

    dev = counts.device
    B = int(shots.shape[0])
    counts = counts.to(torch.int64).contiguous()
    shots = shots.to(torch.int64).contiguous()
    lab = (labels if hasattr(labels, "is_cuda") else torch.from_numpy(np.ascontiguousarray(labels, np.int32))).to(dev, torch.int32).contiguous()
    out = torch.zeros((n_params, 4), dtype=torch.float64, device=dev)
    gd = torch.empty((n_params, B), dtype=torch.float64, device=dev) if want_grad_data else None
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(_lib.lib().vqa_class_stats_dev(dev.index or 0, C.c_void_p(st), p(counts), p(shots), B, int(n_params), p(lab),
                                              int(kind), int(bool(want_variances)), p(out), p(gd)))
    return out, gd
