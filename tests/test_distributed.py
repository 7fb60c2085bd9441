"""Two-rank (gloo, CPU) run of the sharded gradient path against the one-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle_circuit import OracleCircuit
from vqa_amd.distributed import mse_gradient_sums, shard_bounds, sharded_mse_gradient
from vqa_amd.observable import z_observable
from vqa_amd.qml_utils import create_ibm_HEA_ansatz


def problem():
    rng = np.random.default_rng(0)
    n, B = 4, 11
    x = rng.uniform(0, np.pi, (B, n))
    y = np.sin(x.sum(axis=1))
    c = create_ibm_HEA_ansatz(n, 1, circuit_cls=OracleCircuit)
    theta = rng.uniform(0, 2 * np.pi, c.n_theta)
    return n, c, x, y, theta, 1.2


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, c, x, y, theta, w = problem()
    lo, hi = shard_bounds(len(x), rank, world)
    g = sharded_mse_gradient(c, theta, c.upload_inputs(x[lo:hi]), y[lo:hi], w, z_observable(n, 0))
    q.put((rank, g.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 64, 65537):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_two_rank_gradient_equals_single_process():
    n, c, x, y, theta, w = problem()
    e, _ = c.psr_full(theta, c.upload_inputs(x), obs_list=[z_observable(n, 0)])
    want = mse_gradient_sums(e[:, :, 0], y, w) / len(x)
    single = sharded_mse_gradient(c, theta, c.upload_inputs(x), y, w, z_observable(n, 0)).numpy()
    assert np.max(np.abs(single - want)) < 1e-14

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(2):
        assert np.max(np.abs(got[r] - want)) < 1e-13


def vqe_problem():
    from test_vqe_target import make
    return make(n=5, d=2, min_approx=-10)


def vqe_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    v, p = vqe_problem()
    v.group = dist.group.WORLD
    out = v.iEval(p, np.full(v.n_params, 100))
    q.put((rank, out["grad"], out["gvar_list"], v.circuit_counter[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_vqe_gradient_shards_the_parameters():
    """The Pauli-sum target with its shifted evaluations dealt round-robin to two ranks (each sweeps
    its own parameters, one all-reduce) against the one-process gradient."""
    v, p = vqe_problem()
    want = v.iEval(p, np.full(v.n_params, 100))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=vqe_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = [q.get(timeout=120) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for rank, grad, gvar, n_circ in got:
        assert np.max(np.abs(grad - want["grad"])) < 1e-13
        assert np.max(np.abs(gvar - want["gvar_list"])) < 1e-13
        assert n_circ == 2 * v.n_params


def vqe_noisy_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_vqe_target import make
    v, p = make(n=5, d=2, min_approx=0)
    v.group = dist.group.WORLD
    np.random.seed(100 + rank)          # the ranks' generators differ on purpose
    out = v.iEval(p, np.full(v.n_params, 100))
    q.put((rank, out["grad"], out["gvar_list"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_vqe_noisy_gradients_agree():
    """CLT mode with the parameters sharded: every rank must return the SAME noisy gradient (rank 0's
    deviates are broadcast), equal to what one process seeded like rank 0 computes."""
    from test_vqe_target import make
    v, p = make(n=5, d=2, min_approx=0)
    np.random.seed(100)
    want = v.iEval(p, np.full(v.n_params, 100))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=vqe_noisy_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = sorted([q.get(timeout=120) for _ in procs], key=lambda g: g[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert np.array_equal(got[0][1], got[1][1]) and np.array_equal(got[0][2], got[1][2])
    assert np.max(np.abs(got[0][1] - want["grad"])) < 1e-12
