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
        # the library's estimator epilogue: one pass over the matrix, fixed summation order; the sums stay
        # on the device for the all-reduce
        from .stats import GD, mse_stats

        sums = mse_stats(e, y_t, w, device_sums=True)[0][:, GD].contiguous()
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


class BatchShard:
    """How a target object with a process group splits one mini-batch: every rank holds the whole data
    set, rank r evaluates the contiguous block [lo, hi) of the batch's samples, and the ranks add the
    batch sums of their blocks with ONE all-reduce (reduce()).  The batch itself -- which samples, and
    the seed of the device-side random streams -- is rank 0's: bcast_ints() hands it to the others, so
    the ranks' NumPy generators need not agree."""

    def __init__(self, group, n_items, device=None):
        import torch
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n = int(n_items)
        self.lo, self.hi = shard_bounds(self.n, self.rank, self.world)
        # NCCL / RCCL moves device tensors only; gloo takes host tensors
        self.device = torch.device(device if (device is not None and dist.get_backend(group) != "gloo") else "cpu")

    @staticmethod
    def of(target, n_items):
        """The shard of `target` (an object with a .group attribute) for a batch of n_items, or None when
        the object has no group or torch.distributed is not initialised."""
        group = getattr(target, "group", None)
        if group is None:
            return None
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            return None
        dev = None
        circ = getattr(target, "circuit", None)
        if circ is not None and getattr(circ, "device", None) is not None and hasattr(circ, "handle"):
            dev = f"cuda:{circ.device}"
        return BatchShard(group, n_items, dev)

    def bcast_ints(self, values):
        """Rank 0's integer array on every rank."""
        import torch

        t = torch.as_tensor(np.asarray(values, dtype=np.int64)).to(self.device)
        self.dist.broadcast(t, src=self.dist.get_global_rank(self.group, 0) if hasattr(self.dist, "get_global_rank") else 0,
                            group=self.group)
        return t.cpu().numpy()

    def reduce(self, sums, per_item=None):
        """All-reduce of the batch sums (any shape) and, in the same buffer, of a per-item vector that every
        rank fills for its own block (the unshifted estimates of its samples).  sums / per_item: torch
        tensors on any device or numpy arrays.  Returns (sums, per-item vector of all n items) as numpy."""
        import torch

        as_t = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, np.float64))
        s = as_t(sums).to(torch.float64)
        dev = s.device if s.is_cuda else self.device
        buf = torch.zeros(s.numel() + (self.n if per_item is not None else 0), dtype=torch.float64, device=dev)
        buf[:s.numel()] = s.reshape(-1).to(dev)
        if per_item is not None:
            buf[s.numel() + self.lo:s.numel() + self.hi] = as_t(per_item).to(torch.float64).reshape(-1).to(dev)
        all_reduce_sum(buf, group=self.group)
        h = buf.cpu().numpy()
        return h[:s.numel()].reshape(tuple(s.shape)), (h[s.numel():] if per_item is not None else None)
