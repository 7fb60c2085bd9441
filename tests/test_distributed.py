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


# ---- the target objects with a process group: the mini-batch of a gradient oracle in blocks over the ranks ----
def target_problem(min_approx):
    from test_host_layer import small_regressor
    q, p = small_regressor(min_approx=min_approx, n=4, depth=1, n_train=13)
    return q, p


def given_estimates(q, p, eval_ind):
    """A deterministic matrix of 'estimates' for the sweep of a batch: the exact values, perturbed."""
    E = q._psr_all(p, eval_ind)
    rng = np.random.default_rng(5)
    return np.clip(E + 0.05 * rng.standard_normal(E.shape), -0.999, 0.999)


def target_calls(q, p, shift=0):
    """(exact-mode gradient, MSE_iEval with the statistics reduced where the expectation values are, counters):
    what a rank of the group or a single process computes."""
    from oracle import stats_numpy

    shots = np.full(q.n_params, 40)
    np.random.seed(11 + shift)
    q.remain_ind = []
    g = q.MSE_grad_eval(p, shots, mini_batch_size=9, option_min_approx=-10)
    q._stats_fn = stats_numpy.mse_stats
    np.random.seed(12 + shift)
    q.remain_ind = []
    mn, wt, ind, y = q._batch_setup(9, 0, None)
    out = q._ieval_on_device(p, shots, wt, ind, y, 0, 9, True, True, _estimates=given_estimates(q, p, ind))
    counters = (q.shots_counter[0], q.call_counter[0], q.circuit_counter[0], q.epoch_counter[0])
    return g, out, counters, q.cache_train_pred.copy()


def target_worker(rank, world, port, qu):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q, p = target_problem(0)
    q.group = dist.group.WORLD
    # the ranks' NumPy generators differ on purpose: the batch (and the seed of the random streams) is rank 0's
    g, out, counters, pred = target_calls(q, p, shift=100 * rank)
    out = {k: np.asarray(v) for k, v in out.items() if k != "grad_data_list"}
    qu.put((rank, g, out, counters, pred))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_target_object_equals_single_process():
    """MSE_grad_eval (exact mode) and the device-side form of MSE_iEval of a regressor whose mini-batch is split
    over two ranks: every rank returns what one process returns, to rounding, with the same counters."""
    q, p = target_problem(0)
    want_g, want_out, want_counters, want_pred = target_calls(q, p)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    qu = ctx.Queue()
    procs = [ctx.Process(target=target_worker, args=(r, 2, port, qu)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = [qu.get(timeout=180) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for rank, g, out, counters, pred in got:
        assert np.max(np.abs(g - want_g)) < 1e-13
        for k in ("grad", "gvar_list_EV", "gvar_list_EVi", "gvar_list_EV0", "gvar_list_b"):
            assert np.max(np.abs(out[k] - want_out[k])) < 1e-13 * max(1.0, np.max(np.abs(want_out[k]))), k
        assert counters == want_counters
        assert np.max(np.abs(pred - want_pred)) < 1e-13
