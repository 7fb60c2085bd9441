"""Generates trajectories.json: parameter trajectories and shot totals produced by
the REFERENCE's optimizers (vqa.VQA_optimizers.qml_optimizers, imported from
/root/reference) driving this package's QNN regressor on the CPU-oracle
circuit, in CLT mode (min_approx_shots=0) with a fixed NumPy seed.

    cd /tmp && PYTHONPATH=/root/reference python /root/repo/tests/golden/make_trajectories.py

The reference writes log files into the working directory, hence /tmp.  The
reference's own QNN classes need qulacs / skqulacs, which are not installed,
so the target object is this package's; what the vectors pin is the
optimizer side (step rules, rule timing, counters, parameter wrapping).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from oracle_circuit import OracleCircuit  # noqa: E402
from vqa.VQA_optimizers import qml_optimizers as ref  # noqa: E402
from vqa_amd.qml_utils import create_ibm_HEA_ansatz  # noqa: E402
from vqa_amd.qnn_base import myQNN_base  # noqa: E402

CASES = {
    "Adam": dict(LR=0.05, shots=100, batch=8, iters=12),
    "SGD": dict(LR=0.1, shots=64, batch=5, iters=12),
    "SGDM": dict(LR=0.02, shots=50, batch=7, iters=12),
}


def dataset(seed=7, n_qubit=4, depth=1, n_train=24, n_test=10):
    rng = np.random.default_rng(seed)
    circuit = create_ibm_HEA_ansatz(n_qubit, depth, circuit_cls=OracleCircuit)
    qnn = myQNN_base(n_qubit, circuit, min_approx_shots=0)
    x = rng.uniform(0, 1, (n_train + n_test, n_qubit))
    target = np.concatenate([rng.uniform(-np.pi, np.pi, qnn.n_c_params), [1.3]])
    y = qnn.generate_labels(x, target)
    init = np.concatenate([rng.uniform(-np.pi, np.pi, qnn.n_c_params), [1.0]])
    return dict(n_qubit=n_qubit, depth=depth, x_train=x[:n_train], y_train=y[:n_train], x_test=x[n_train:],
                y_test=y[n_train:], init=init)


def hpara(case, n_params):
    return dict(LR_rule=lambda t, v=case["LR"]: v / (1 + 0.1 * t),
                s_list_rule=lambda t, v=case["shots"]: np.full(n_params, v + 2.0 * t),
                batch_size_rule=lambda t, v=case["batch"]: v + (t % 3),
                LR_annealing_by_shots=False, s_list_annealing_by_shots=False, batch_size_annealing_by_shots=False)  # (n_shots stays shot-indexed: the reference needs one such rule to exist)


def main():
    d = dataset()
    out = {"dataset": {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in d.items()}, "cases": {}}
    for name, case in CASES.items():
        circuit = create_ibm_HEA_ansatz(d["n_qubit"], d["depth"], circuit_cls=OracleCircuit)
        qnn = myQNN_base(d["n_qubit"], circuit, d["x_train"], d["y_train"], d["x_test"], d["y_test"], min_approx_shots=0)
        qnn.set_methods_for_optimizer(myQNN_base._LOSS_METHODS["mean_squared_error"])
        np.random.seed(1234)
        opt = getattr(ref, name)(hpara=hpara(case, qnn.n_params),
                                 init_spec_dict={"target_obj": qnn, "init_params": d["init"].copy()})
        res = opt.optimize(options=dict(max_iter=case["iters"], track_var_names=[], track_indicator="iteration_num",
                                        take_SA=False, take_BMA=False, Monitor_iter=False))
        out["cases"][name] = dict(
            case, seed=1234,
            para_evol=[np.asarray(p).tolist() for p in res["para_evol"]],
            func_evol=[{k: float(v) for k, v in f.items()} for f in res["func_evol"]],
            total_shots=[int(v) for v in res["total_shots"]],
            total_call=[int(v) for v in res["total_call"]],
            total_epoch=[int(v) for v in res["total_epoch"]],
            total_time=[float(v) for v in res["total_time"]],
        )
    with open(os.path.join(HERE, "trajectories.json"), "w") as f:
        json.dump(out, f)
    print("wrote", os.path.join(HERE, "trajectories.json"))


if __name__ == "__main__":
    main()
