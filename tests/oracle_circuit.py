"""Circuit object with the batched interface of vqa_amd.LearningCircuit, computed
by the CPU oracle.  Test infrastructure: it lets the host layer (QNN target
objects, optimizers) run on machines without a GPU and gives the GPU tests
the numbers to compare with."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from oracle import cpu_oracle  # noqa: E402
from vqa_amd.circuit import GateList  # noqa: E402


class OracleCircuit(GateList):
    n_threads = 4

    def upload_inputs(self, x_list):
        return self.input_angles(x_list)

    def eval_batch(self, theta, inputs, sample_idx=None, shift_param=None, shift_val=None, obs_list=(),
                   n_marg_bits=0, **unused):
        obs = [o.arrays() for o in obs_list]
        return cpu_oracle.eval_batch(self, theta, inputs, sample_idx, shift_param, shift_val, obs, n_marg_bits,
                                     self.n_threads)

    def psr_full(self, theta, inputs, sel=None, n_sel=None, shift=0.5 * np.pi, obs_list=(), n_marg_bits=0,
                 params=None, **unused):
        B = 0 if inputs is None else inputs.shape[0]
        if n_sel is None:
            n_sel = len(sel) if sel is not None else max(B, 1)
        K = 2 * self.n_theta + 1
        si, sp, sv = cpu_oracle.psr_lists(n_sel, self.n_theta, shift, sel)
        keep = np.ones(len(si), bool) if params is None else ((sp < 0) | np.isin(sp, np.asarray(params)))
        e = np.zeros((len(si), len(obs_list)))
        m = np.zeros((len(si), 1 << n_marg_bits)) if n_marg_bits else None
        ek, mk = self.eval_batch(theta, inputs, si[keep], sp[keep], sv[keep], obs_list, n_marg_bits)
        e[keep] = ek
        if m is not None:
            m[keep] = mk
        return e.reshape(n_sel, K, len(obs_list)), None if m is None else m.reshape(n_sel, K, -1)
