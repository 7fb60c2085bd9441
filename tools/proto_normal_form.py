"""Prototype (NumPy) of the register-blocked engine's circuit normal form:
alternating U-stages (one 2x2 per qubit) and D-stages (diagonal), with every
2x2 re-expressed as RZ(alpha) RY(beta) RZ(gamma) so that the dense part is a
real rotation and all phases are merged into the neighbouring D-stages.
Development aid for vqa_amd/csrc/reg_compile.cpp; checks probabilities against
the dense oracle.  Not part of the product or of the test suite."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vqa_amd.circuit import GateList
from vqa_amd.qml_utils import create_ibm_HEA_ansatz, create_jerbi_ansatz, create_ry_rz_cz_ansatz
from oracle import dense_numpy, cpu_oracle
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import random_circuit

H, X, Y, Z, RX, RY, RZ, CNOT, CZ = range(9)


class UStage:
    def __init__(self): self.g = {}          # qubit -> list of op indices (time order)
class DStage:
    def __init__(self): self.rz = {}; self.zz = {}; self.cz = []; self.z = []   # rz: q -> [op idx]; zz: (a,b) -> [op idx]


def normal_form(ops, n):
    """ops: structured array. returns list of stages."""
    # expand lone CNOTs, detect CNOT RZ CNOT sandwiches
    items = []  # ('g1', kind, q, opidx) | ('zz', a, b, opidx) | ('cz', a, b)
    i = 0
    N = len(ops)
    while i < N:
        o = ops[i]
        k = int(o['kind'])
        if k == CNOT:
            if i + 2 < N and int(ops[i+1]['kind']) == RZ and int(ops[i+1]['q0']) == int(o['q0']) \
               and int(ops[i+2]['kind']) == CNOT and int(ops[i+2]['q0']) == int(o['q0']) and int(ops[i+2]['q1']) == int(o['q1']):
                items.append(('zz', int(o['q1']), int(o['q0']), i + 1)); i += 3; continue
            items.append(('g1', H, int(o['q0']), -1)); items.append(('cz', int(o['q0']), int(o['q1']))); items.append(('g1', H, int(o['q0']), -1))
        elif k == CZ:
            items.append(('cz', int(o['q0']), int(o['q1'])))
        else:
            items.append(('g1', k, int(o['q0']), i))
        i += 1
    stages = []
    def touches(d, q):
        return q in d.rz or q in d.z or any(q in p for p in d.zz) or any(q in p for p in d.cz)
    for it in items:
        last = stages[-1] if stages else None
        if it[0] == 'g1':
            _, k, q, idx = it
            diag = k in (RZ, Z)
            if isinstance(last, UStage):
                last.g.setdefault(q, []).append((k, idx))
            elif isinstance(last, DStage):
                if diag:
                    if k == RZ: last.rz.setdefault(q, []).append(idx)
                    else: last.z.append(q)
                elif not touches(last, q) and len(stages) >= 2:
                    stages[-2].g.setdefault(q, []).append((k, idx))
                else:
                    u = UStage(); u.g[q] = [(k, idx)]; stages.append(u)
            else:
                u = UStage(); u.g[q] = [(k, idx)]; stages.append(u)
        else:
            if not isinstance(last, DStage):
                last = DStage(); stages.append(last)
            if it[0] == 'zz':
                a, b = sorted((it[1], it[2])); last.zz.setdefault((a, b), []).append(it[3])
            else:
                last.cz.append((it[1], it[2]))
    return stages


def angle(o, theta, row, sp, sv):
    v = 0.0
    if o['src'] == 1: v = theta[o['idx']] + (sv if o['idx'] == sp else 0)
    elif o['src'] == 2: v = row[o['idx']]
    return o['coef'] * v + o['offset']


def mat1(k, a):
    c, s = np.cos(a / 2), np.sin(a / 2)
    return {H: np.array([[1, 1], [1, -1]]) / np.sqrt(2), X: np.array([[0, 1], [1, 0]]), Y: np.array([[0, -1j], [1j, 0]]),
            Z: np.diag([1, -1]), RX: np.array([[c, 1j * s], [1j * s, c]]), RY: np.array([[c, s], [-s, c]]),
            RZ: np.diag([c + 1j * s, c - 1j * s])}[k].astype(complex)


def sqrt_ph(z):
    if z.real >= 0:
        w = 1 + z
    else:
        w = 1j * (1 - z)
    return w / abs(w)


def zyz(U):
    """U = e^{i d} RZ(al) RY(be) RZ(ga); returns ca, sa, A=e^{i al/2}, G=e^{i ga/2}."""
    u00, u01, u10, u11 = U[0, 0], U[0, 1], U[1, 0], U[1, 1]
    ca, sa = abs(u00), abs(u01)
    nrm = np.hypot(ca, sa); ca, sa = ca / nrm, sa / nrm
    det = u00 * u11 - u01 * u10
    eid = sqrt_ph(det / abs(det))
    tiny = 1e-150
    Xp = (u00 / abs(u00)) * np.conj(eid) if abs(u00) > tiny else 1.0 + 0j
    Yp = (u01 / abs(u01)) * np.conj(eid) if abs(u01) > tiny else 1.0 + 0j
    if abs(u00) > tiny and abs(u01) > tiny:
        A = sqrt_ph(Xp * Yp); G = Xp * np.conj(A)
    elif abs(u01) <= tiny:
        A, G = Xp, 1.0 + 0j
    else:
        A, G = Yp, 1.0 + 0j
    return ca, sa, A, G


def run_normal_form(c, theta, row, sp=-1, sv=0.0):
    n = c.n_qubit
    ops = c.op_table()
    stages = normal_form(ops, n)
    D = 1 << n
    # per U-stage: ry (ca, sa), A, G per qubit
    udata = []
    for s in stages:
        if isinstance(s, UStage):
            d = {}
            for q, lst in s.g.items():
                M = np.eye(2, dtype=complex)
                for k, idx in lst:
                    M = mat1(k, angle(ops[idx], theta, row, sp, sv) if idx >= 0 else 0.0) @ M
                d[q] = zyz(M)
            udata.append(d)
        else:
            udata.append(None)
    psi = np.zeros(D, complex); psi[0] = 1
    k = np.arange(D)
    bit = lambda q: (k >> q) & 1
    # make sure there is a D-stage between/around U-stages to absorb phases: emulate with explicit phase application
    for si, s in enumerate(stages):
        if isinstance(s, UStage):
            for q, (ca, sa, A, G) in udata[si].items():
                # pre phase (gamma), RY, post phase (alpha) -- applied here directly; the engine merges them into D-stages
                psi = psi * np.where(bit(q) == 0, G, np.conj(G))
                p0 = psi.copy()
                partner = p0[k ^ (1 << q)]
                psi = np.where(bit(q) == 0, ca * p0 + sa * partner, ca * p0 - sa * partner)
                psi = psi * np.where(bit(q) == 0, A, np.conj(A))
        else:
            for q, lst in s.rz.items():
                a = sum(angle(ops[i], theta, row, sp, sv) for i in lst)
                ph = np.exp(0.5j * a); psi = psi * np.where(bit(q) == 0, ph, np.conj(ph))
            for (a_, b_), lst in s.zz.items():
                a = sum(angle(ops[i], theta, row, sp, sv) for i in lst)
                ph = np.exp(0.5j * a); psi = psi * np.where(bit(a_) == bit(b_), ph, np.conj(ph))
            for (a_, b_) in s.cz:
                psi = psi * np.where((bit(a_) & bit(b_)) == 1, -1, 1)
            for q in s.z:
                psi = psi * np.where(bit(q) == 1, -1, 1)
    return psi, stages


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for c in [create_ibm_HEA_ansatz(5, 2, circuit_cls=GateList), create_jerbi_ansatz(4, 2, circuit_cls=GateList),
              create_ry_rz_cz_ansatz(4, 3, circuit_cls=GateList), random_circuit(GateList, 5, 80, 1), random_circuit(GateList, 6, 120, 2),
              random_circuit(GateList, 3, 200, 3)]:
        theta = rng.uniform(-3, 3, c.n_theta)
        x = rng.uniform(0, 3, max(c.n_qubit, 3))
        row = c.input_row(x)
        sp = int(rng.integers(max(c.n_theta, 1))) if c.n_theta else -1
        psi, stages = run_normal_form(c, theta, row, sp, 0.5 * np.pi)
        ref = cpu_oracle.statevector(c, theta, row, sp, 0.5 * np.pi)
        nu = sum(isinstance(s, UStage) for s in stages)
        print(c.n_qubit, len(c.op_table()), "stages", len(stages), "U", nu, "prob err", np.abs(np.abs(psi) ** 2 - np.abs(ref) ** 2).max(),
              "overlap", abs(np.vdot(psi, ref)))
