"""Sharding of the parameter-shift hot path over GPUs: one process per GPU,
every rank evaluates the full sweep for its own slice of the mini-batch and
the ranks exchange only the P+1 gradient sums (one all-reduce; RCCL on GPUs,
gloo on CPUs).  Circuit evaluations of different samples are independent
(vqa/QNN/myQNNs_base.py:1381-1394 loops over them), so there is no other
exchange on the path.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_items: int, rank: int, world: int):
    """Contiguous, balanced slice [lo, hi) of n_items for `rank` of `world`."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def mse_gradient_sums(e, y, w):
    """Per-rank sums behind the regressor's exact MSE gradient
    (vqa/QNN/myQNNs_base.py:1378, 1393): e is [B, 2P+1] with columns
    (base, +shift_0, -shift_0, ...); returns the P+1 sums over the B samples.
    Works on numpy arrays and on torch tensors (any device)."""
    e0 = e[:, 0]
    resid = (w * e0 - y) * w
    circ = (resid[:, None] * (e[:, 1::2] - e[:, 2::2])).sum(0)
    scale = (2.0 * (-y * e0 + w * e0 * e0)).sum()
    if isinstance(e, np.ndarray):
        return np.concatenate([circ, [scale]])
    import torch

    return torch.cat([circ, scale.reshape(1)])


def all_reduce_sum(t, group=None):
    """In-place sum over the ranks of `group`.  RCCL reduces device tensors where they are; the gloo
    backend (CPU rehearsals and tests) gets a host copy: its path for device tensors stages through
    pinned buffers per call and is slower than the kernel it follows by orders of magnitude."""
    import torch.distributed as dist

    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)
    return t


def sharded_mse_gradient(circuit, theta, inputs, y, w, obs, n_global=None, group=None, out_expect=None):
    """Exact MSE gradient of the regressor over a mini-batch that is sharded across
    the ranks of `group`: `inputs` / `y` are THIS rank's samples.  Returns the
    gradient of the global mean (identical on every rank) as a torch tensor on
    the device the circuit computes on."""
    import torch
    import torch.distributed as dist

    e, _ = circuit.psr_full(theta, inputs, n_sel=int(inputs.shape[0]), obs_list=[obs], out_expect=out_expect)
    e = e[:, :, 0]
    if isinstance(e, np.ndarray):
        e = torch.from_numpy(e)
    y_t = y if isinstance(y, torch.Tensor) else torch.from_numpy(np.asarray(y, np.float64)).to(e.device)
    if e.is_cuda:
        # the library's estimator epilogue: one pass over the matrix, fixed summation order
        from .stats import GD, mse_stats

        sums = torch.from_numpy(np.ascontiguousarray(mse_stats(e, y_t, w)[0][:, GD])).to(e.device)
    else:
        sums = mse_gradient_sums(e, y_t, w)
    distributed = dist.is_available() and dist.is_initialized()
    if n_global is None:
        count = torch.tensor([float(e.shape[0])], dtype=torch.float64, device=e.device)
        if distributed:
            all_reduce_sum(count, group=group)
        n_global = float(count.item())
    if distributed:
        all_reduce_sum(sums, group=group)
    return sums / n_global
