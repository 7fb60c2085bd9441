"""Generates call_traces.json: every call the REFERENCE's shot-adaptive optimizers
(vqa.VQA_optimizers.qml_optimizers.SantaQlaus / gCANS / SPSA, imported from /root/reference)
make into this package's QNN regressor -- which oracle ('iEvaluate', 'iEvaluate_iRandom',
'loss', 'func_to_track'), with which positional and keyword arguments -- and what came back,
on the CPU-oracle circuit.

    cd /tmp && PYTHONPATH=/root/reference python /root/repo/tests/golden/make_call_traces.py

Every recorded call re-seeds NumPy's global generator with seed0 + its call number first, so
that a replay (tests/test_call_traces.py) needs no optimizer: the same calls in the same
order on a fresh target object must give the same numbers.  What the vectors pin: the
optimizer <-> target protocol as the reference really drives it (argument order, keywords,
shot lists it derives from earlier answers, mini-batch sizes, the multinomial allocation of
gCANS), and, on the GPU, the HIP path under those call patterns.  (Adam_ASS is left out: it
fails inside the reference before its first call, qml_optimizers.py:789-812 never sets s_list.)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from make_trajectories import dataset  # noqa: E402
from oracle_circuit import OracleCircuit  # noqa: E402
from vqa.VQA_optimizers import qml_optimizers as ref  # noqa: E402
from vqa_amd.qml_utils import create_ibm_HEA_ansatz  # noqa: E402
from vqa_amd.qnn_base import myQNN_base  # noqa: E402

SEED0 = 4000
ITERS = {"SantaQlaus": 6, "gCANS": 6, "SPSA": 6}


def enc(v):
    if isinstance(v, np.ndarray):
        return {"nd": v.tolist(), "dtype": str(v.dtype)}
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return float(v)
    if isinstance(v, dict):
        return {"dict": {k: enc(x) for k, x in v.items() if k != "grad_data_list"}}
    if isinstance(v, (list, tuple)):
        return {"list": [enc(x) for x in v]}
    return v


def hpara(qnn):
    return dict(LR_annealing_by_shots=False, beta_annealing_by_shots=False, G1_scaling_annealing_by_shots=False,
                c_t_annealing_by_shots=False, batch_size_annealing_by_shots=False, s_list_annealing_by_shots=False,
                LR_rule=lambda t: 0.05, n_shots_rule=lambda t: 50, batch_size_rule=lambda t: 6,
                s_list_rule=lambda t: np.full(qnn.n_params, 60))


def main():
    d = dataset()
    out = {"seed0": SEED0, "cases": {}}
    for name, iters in ITERS.items():
        circuit = create_ibm_HEA_ansatz(d["n_qubit"], d["depth"], circuit_cls=OracleCircuit)
        qnn = myQNN_base(d["n_qubit"], circuit, d["x_train"], d["y_train"], d["x_test"], d["y_test"], min_approx_shots=0)
        qnn.set_methods_for_optimizer(myQNN_base._LOSS_METHODS["mean_squared_error"])
        calls = []

        def recorder(key, fn):
            def wrapped(*args, **kwargs):
                np.random.seed(SEED0 + len(calls))
                before = [int(qnn.shots_counter[0]), int(qnn.call_counter[0]), int(qnn.circuit_counter[0]),
                          int(qnn.epoch_counter[0])]
                res = fn(*args, **kwargs)
                calls.append({"method": key, "args": [enc(a) for a in args], "kwargs": {k: enc(v) for k, v in kwargs.items()},
                              "out": enc(res),
                              # what the call added to the shot / call / circuit / epoch counters (the optimizer
                              # resets them between steps)
                              "counters": [int(a[0]) - b for a, b in zip((qnn.shots_counter, qnn.call_counter,
                                                                          qnn.circuit_counter, qnn.epoch_counter), before)]})
                return res
            return wrapped

        for key in list(qnn.methods_for_optimization):
            qnn.methods_for_optimization[key] = recorder(key, qnn.methods_for_optimization[key])
        np.random.seed(1)
        opt = getattr(ref, name)(hpara=hpara(qnn), init_spec_dict={"target_obj": qnn, "init_params": d["init"].copy()})
        opt.optimize(options=dict(max_iter=iters, max_shots=200000, track_var_names=[], track_indicator="iteration_num",
                                  take_SA=False, take_BMA=False, Monitor_iter=False))
        out["cases"][name] = calls
        print(name, len(calls), "calls:", sorted(set(c["method"] for c in calls)))
    with open(os.path.join(HERE, "call_traces.json"), "w") as f:
        json.dump(out, f)
    print("wrote", os.path.join(HERE, "call_traces.json"), os.path.getsize(os.path.join(HERE, "call_traces.json")), "bytes")


if __name__ == "__main__":
    main()
