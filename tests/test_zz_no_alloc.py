"""vqa_psr_full_noalloc_dev -- the sweep entry point whose working memory is a static device
array instead of the library's workspace -- against vqa_psr_full_dev, through the C ABI, on two
calls in a row, the second with a smaller input (the array keeps the first call's contents).
Outputs must be bit-identical to the allocating entry point's where both launch the same
kernels on the same grid, and within 1e-12 where the fixed array makes the grid smaller
(workgroups then take samples in another order; each instance is still computed alike)."""
import ctypes as C

import numpy as np
import pytest

from vqa_amd import _lib
from vqa_amd.observable import Observable
from vqa_amd.qml_utils import create_ibm_HEA_ansatz, create_jerbi_ansatz

pytestmark = pytest.mark.gpu


def sweep(entry, c, theta, x, sel, obs, n_marg_bits, check_free=None):
    import torch

    dev = f"cuda:{c.device}"
    K, n_obs, M = 2 * c.n_theta + 1, len(obs), 1 << n_marg_bits
    n_sel = len(sel) if sel is not None else x.shape[0]
    th = torch.from_numpy(np.ascontiguousarray(theta)).to(dev)
    se = None if sel is None else torch.from_numpy(np.asarray(sel, np.int32)).to(dev)
    oe = torch.full((n_sel, K, n_obs), np.nan, dtype=torch.float64, device=dev)
    om = torch.full((n_sel, K, M), np.nan, dtype=torch.float64, device=dev) if n_marg_bits else None
    oarr = c._obs_array(obs)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    st = torch.cuda.current_stream(c.device).cuda_stream
    _lib.check(getattr(_lib.lib(), entry)(c.handle(), c.device, C.c_void_p(st), p(th), p(x), int(x.shape[0]), p(se), n_sel,
                                           0.5 * np.pi, oarr, n_obs, n_marg_bits, p(oe), p(om)))
    torch.cuda.synchronize()
    if check_free is not None:
        # (measured right behind the call, before torch copies the outputs anywhere)
        assert torch.cuda.mem_get_info(c.device)[0] == check_free, "the allocation-free entry point changed the free memory"
    return oe.cpu().numpy(), None if om is None else om.cpu().numpy()


def problem(builder, n, depth, B, seed):
    from vqa_amd import LearningCircuit

    c = builder(n, depth, seed=seed, circuit_cls=LearningCircuit)
    rng = np.random.default_rng(seed)
    theta = rng.uniform(0, 2 * np.pi, c.n_theta)
    x = c.upload_inputs(rng.uniform(0, np.pi, (B, n)))
    z0, zz = Observable(n), Observable(n)
    z0.add_operator(1.0, "Z 0")
    zz.add_operator(0.7, f"Z 1 Z {n - 1}")
    zz.add_operator(-0.2, "")
    return c, theta, x, [z0, zz]


@pytest.mark.parametrize("first, second", [
    ((create_ibm_HEA_ansatz, 12, 3, 40), (create_ibm_HEA_ansatz, 10, 2, 9)),     # sweep kernel, then a smaller circuit
    ((create_ibm_HEA_ansatz, 6, 5, 300), (create_jerbi_ansatz, 5, 2, 7)),        # grouped sweeps of small circuits
    ((create_ibm_HEA_ansatz, 11, 4, 64), (create_ibm_HEA_ansatz, 11, 4, 3)),     # same circuit, fewer samples
])
def test_noalloc_sweep_matches_the_allocating_entry_point(first, second):
    for k, (builder, n, depth, B) in enumerate((first, second)):
        c, theta, x, obs = problem(builder, n, depth, B, seed=10 * n + k)
        sel = None if k == 0 else np.arange(B - 1, -1, -1)
        for mb in (2, 0):
            we, wm = sweep("vqa_psr_full_dev", c, theta, x, sel, obs, mb)
            ge, gm = sweep("vqa_psr_full_noalloc_dev", c, theta, x, sel, obs, mb)
            # a repeat of the call (everything it could have created lazily exists now) must leave the
            # device's free memory as it found it: the entry point allocates nothing
            import torch
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info(c.device)[0]
            ge2, _ = sweep("vqa_psr_full_noalloc_dev", c, theta, x, sel, obs, mb, check_free=free0)
            assert np.array_equal(ge2, ge) or np.abs(ge2 - ge).max() < 1e-12
            assert not np.isnan(ge).any()
            assert np.abs(ge - we).max() < 1e-12
            if mb:
                assert np.abs(gm - wm).max() < 1e-12


def test_noalloc_rejects_what_does_not_fit():
    c, theta, x, obs = problem(create_ibm_HEA_ansatz, 15, 1, 2, seed=1)       # streaming engine
    with pytest.raises(_lib.VqaError) as e:
        sweep("vqa_psr_full_noalloc_dev", c, theta, x, None, obs, 0)
    assert e.value.code == _lib.EUNSUPPORTED
    # (a circuit whose state images outgrow the array -- 14 qubits with more than about 120 diagonal
    # stages -- does not fit the register engine's LDS tables either and is turned away the same way)
    c, theta, x, obs = problem(create_ibm_HEA_ansatz, 14, 140, 2, seed=2)
    with pytest.raises(_lib.VqaError) as e:
        sweep("vqa_psr_full_noalloc_dev", c, theta, x, None, obs, 0)
    assert e.value.code == _lib.EUNSUPPORTED
