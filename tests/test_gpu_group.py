"""Target objects with a process group on the GPU: the mini-batch of the device-side gradient oracles is split into
blocks over the ranks and the batch sums meet in one all-reduce.  Two ranks share the one GPU of the box here (gloo:
RCCL needs a GPU per rank); the random streams are named by a sample's number in the whole batch, so every rank must
return exactly what one process returns."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def build(cls_name):
    from test_gpu_host_layer import pair
    from vqa_amd.qml_utils import create_ibm_HEA_ansatz
    from vqa_amd.qnn_base import myQNN_base
    from vqa_amd.qnn_classifier import myQNN_classifier

    if cls_name == "regressor":
        (g, p), _ = pair(create_ibm_HEA_ansatz, 6, 2, myQNN_base, n_train=29)
    else:
        (g, p), _ = pair(create_ibm_HEA_ansatz, 5, 1, myQNN_classifier, n_train=29, class_unbias=True)
    g.device_statistics = True
    return g, p


def calls(g, p, cls_name, shift=0):
    shots = np.full(g.n_params, 500)
    out = {}
    np.random.seed(3 + shift)
    g.remain_ind = []
    if cls_name == "regressor":
        out["exact"] = g.MSE_grad_eval(p, shots, mini_batch_size=17, option_min_approx=-10)
        out["clt"] = g.MSE_grad_eval(p, shots, mini_batch_size=17, option_min_approx=0)
        out["shots"] = g.MSE_grad_eval(p, shots, mini_batch_size=17, option_min_approx=-1)
        ie = g.MSE_iEval(p, shots, mini_batch_size=17, option_min_approx=0)
    else:
        out["mse"] = g.MSE_grad_eval(p, shots, mini_batch_size=17, partial_MSE=True)
        out["ce"] = g.CE_grad_eval(p, shots, mini_batch_size=17)
        ie = g.MSE_iEval(p, shots, mini_batch_size=17, partial_MSE=False)
    for k, v in ie.items():
        if k != "grad_data_list":
            out["ieval_" + k] = np.asarray(v)
    out["counters"] = np.array([g.shots_counter[0], g.call_counter[0], g.circuit_counter[0], g.epoch_counter[0]], float)
    return out


def worker(rank, world, port, cls_name, qu):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g, p = build(cls_name)
    g.group = dist.group.WORLD
    out = calls(g, p, cls_name, shift=50 * rank)       # different NumPy streams: the batch and the seeds are rank 0's
    qu.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cls_name", ["regressor", "classifier"])
def test_two_ranks_on_one_gpu_equal_one_process(cls_name):
    import torch.multiprocessing as mp

    g, p = build(cls_name)
    want = calls(g, p, cls_name)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    qu = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, cls_name, qu)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = [qu.get(timeout=300) for _ in procs]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for rank, out in got:
        assert set(out) == set(want)
        for k in want:
            assert np.max(np.abs(out[k] - want[k])) < 1e-12 * max(1.0, np.max(np.abs(want[k]))), (rank, k)
