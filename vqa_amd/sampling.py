"""Shot sampling on the GPU: multinomial counts of the measured low-bit outcomes.

The reference samples basis states from each simulated state and keeps only
``samples % 2`` / ``samples % 4`` (vqa/QNN/myQNNs_base.py:725-727;
vqa/QNN/myQNNs_classifier.py:613-621, 640-649).  ``sample_counts`` draws those
outcome counts directly from the exact marginals the engines return
(``vqa_sample_counts[_dev]`` of the C ABI; Philox counter-based generator, one
row per thread), so finite-shot gradients need no per-sample host work.
qulacs' private generator cannot be reproduced: agreement is in distribution.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def sample_counts(probs, shots, seed: int, stream_id: int = 0, device: int = 0, row_offset: int = 0):
    """counts[e, :] ~ Multinomial(shots[e], probs[e, :]).  row_offset (device path): the number of the
    first row in a longer list that several ranks share; row e draws what row e + row_offset of one
    call over the whole list would draw.

    numpy inputs: host-pointer path, returns an int64 array.  torch cuda
    tensors: device path, queued on the current torch stream, returns an int64
    tensor on the same device."""
    L = _lib.lib()
    seed, stream_id = int(seed) & (2 ** 64 - 1), int(stream_id) & (2 ** 64 - 1)
    if isinstance(probs, np.ndarray) or not hasattr(probs, "is_cuda"):
        p = np.ascontiguousarray(probs, np.float64)
        if p.ndim != 2:
            raise ValueError("probs must be [E, n_outcomes]")
        n = np.ascontiguousarray(np.broadcast_to(np.asarray(shots, np.int64), (p.shape[0],)))
        out = np.zeros(p.shape, np.int64)
        _lib.check(L.vqa_sample_counts(device, _lib.nptr(p), p.shape[1], _lib.nptr(n), p.shape[0], seed, stream_id,
                                       _lib.nptr(out)))
        return out
    import torch

    if probs.dim() != 2:
        raise ValueError("probs must be [E, n_outcomes]")
    dev = probs.device
    p = probs.to(torch.float64).contiguous()
    n = torch.as_tensor(shots, device=dev).to(torch.int64).expand(p.shape[0]).contiguous()
    out = torch.empty(p.shape, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    ptr = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(L.vqa_sample_counts_shard_dev(dev.index or 0, C.c_void_p(st), ptr(p), p.shape[1], ptr(n), p.shape[0],
                                             seed, stream_id, ptr(out), int(row_offset)))
    return out


def sample_z_means(expect, shots, seed: int, stream_id: int = 0, device: int = 0):
    """Sample means of +-1 outcomes with exact expectation `expect` (any shape) from `shots`
    shots each: (n - 2 k) / n with k ~ Binomial(n, (1 - E) / 2)
    (vqa/QNN/myQNNs_base.py:725-733 for every element at once)."""
    if isinstance(expect, np.ndarray) or not hasattr(expect, "is_cuda"):
        e = np.asarray(expect, np.float64)
        p_minus = np.clip(0.5 * (1.0 - e.reshape(-1)), 0.0, 1.0)
        n = np.broadcast_to(np.asarray(shots, np.int64), e.shape).reshape(-1)
        k = sample_counts(np.stack([p_minus, 1.0 - p_minus], axis=1), n, seed, stream_id, device)[:, 0]
        return ((n - 2.0 * k) / n).reshape(e.shape)
    import torch

    e = expect.to(torch.float64)
    p_minus = (0.5 * (1.0 - e.reshape(-1))).clamp(0.0, 1.0)
    n = torch.as_tensor(shots, device=e.device).to(torch.int64).expand(e.shape).reshape(-1)
    k = sample_counts(torch.stack([p_minus, 1.0 - p_minus], dim=1), n, seed, stream_id)[:, 0]
    nf = n.to(torch.float64)
    return ((nf - 2.0 * k.to(torch.float64)) / nf).reshape(e.shape)
