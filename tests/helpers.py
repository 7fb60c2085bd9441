"""Shared generators for the parity tests."""
import numpy as np

from vqa_amd.observable import Observable

TOL = 1e-10  # absolute fp64 tolerance on expectation values / probabilities


def random_circuit(cls, n, n_gates, seed, n_theta=5, n_input=3):
    """Random gate list over every gate kind, with trainable, shared, input and
    constant rotations."""
    rng = np.random.default_rng(seed)
    c = cls(n)
    created = 0
    # make sure the parameter / input tables have their full size first
    for i in range(n_theta):
        getattr(c, f"add_parametric_R{'XYZ'[i % 3]}_gate")(int(rng.integers(n)), float(rng.uniform(0, 6.28)))
    for i in range(n_input):
        getattr(c, f"add_input_R{'ZYX'[i % 3]}_gate")(int(rng.integers(n)), lambda x, i=i: x[i % len(x)] * (1 + 0.1 * i))
    for _ in range(n_gates):
        k = int(rng.integers(12))
        q = int(rng.integers(n))
        if k == 0: c.add_H_gate(q)
        elif k == 1: c.add_X_gate(q)
        elif k == 2: c.add_Y_gate(q)
        elif k == 3: c.add_Z_gate(q)
        elif k in (4, 5, 6):
            ax = "XYZ"[k - 4]
            mode = int(rng.integers(3))
            if mode == 0 or n_theta == 0:
                getattr(c, f"add_R{ax}_gate")(q, float(rng.uniform(-3, 3)))
            elif mode == 1:
                getattr(c, f"add_parametric_R{ax}_gate")(q, 0.0, share_with=int(rng.integers(n_theta)),
                                                          share_with_coef=float(rng.uniform(-2, 2)))
            else:
                getattr(c, f"add_R{ax}_gate")(q, float(rng.uniform(-3, 3)))
        elif n > 1 and k in (7, 8, 9):
            q1 = int((q + 1 + rng.integers(n - 1)) % n)
            c.add_CNOT_gate(q, q1)
        elif n > 1:
            q1 = int((q + 1 + rng.integers(n - 1)) % n)
            c.add_CZ_gate(q, q1)
        else:
            c.add_H_gate(q)
        created += 1
    return c


def random_sweep_circuit(cls, n, n_gates, seed, n_theta=24, n_input=3):
    """Random gate list in which every trainable parameter drives exactly one rotation (what a
    parameter-shift sweep with one moved state per parameter takes), at random places among constant
    gates of every kind, input rotations, lone CNOTs, CZs and CNOT - RZ - CNOT sandwiches."""
    rng = np.random.default_rng(seed)
    c = cls(n)
    where = set(int(v) for v in rng.choice(n_gates, n_theta, replace=False))
    for i in range(n_input):
        getattr(c, f"add_input_R{'ZYX'[i % 3]}_gate")(int(rng.integers(n)), lambda x, i=i: x[i % len(x)] * (1 + 0.1 * i))
    for g in range(n_gates):
        q = int(rng.integers(n))
        if g in where:
            getattr(c, f"add_parametric_R{'XYZ'[int(rng.integers(3))]}_gate")(q, float(rng.uniform(0, 6.28)))
            continue
        k = int(rng.integers(12))
        q1 = int((q + 1 + rng.integers(n - 1)) % n)
        if k == 0: c.add_H_gate(q)
        elif k == 1: getattr(c, f"add_{'XYZ'[int(rng.integers(3))]}_gate")(q)
        elif k in (2, 3, 4): getattr(c, f"add_R{'XYZ'[k - 2]}_gate")(q, float(rng.uniform(-3, 3)))
        elif k == 5: getattr(c, f"add_input_R{'XYZ'[int(rng.integers(3))]}_gate")(q, lambda x, q=q: x[q % len(x)] - 0.2)
        elif k in (6, 7): c.add_CNOT_gate(q, q1)
        elif k in (8, 9): c.add_CZ_gate(q, q1)
        else:
            c.add_CNOT_gate(q, q1)
            c.add_input_RZ_gate(q1, lambda x, q=q, q1=q1: x[q % len(x)] * x[q1 % len(x)])
            c.add_CNOT_gate(q, q1)
    return c


def random_observables(n, seed, n_obs=3, max_terms=4, xy=False):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_obs):
        o = Observable(n)
        for _ in range(int(rng.integers(1, max_terms + 1))):
            toks = []
            for q in range(n):
                r = rng.random()
                if r < 0.25: toks.append(f"Z {q}")
                elif xy and r < 0.4: toks.append(f"X {q}")
                elif xy and r < 0.5: toks.append(f"Y {q}")
            o.add_operator(float(rng.uniform(-1, 1)), " ".join(toks))
        out.append(o)
    return out


def obs_tuples(obs_list):
    return [o.arrays() for o in obs_list]
