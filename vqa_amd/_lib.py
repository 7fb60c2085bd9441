"""ctypes binding of vqa_amd/csrc/libvqahip.so (the C ABI of include/vqa_hip.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded
the first call raises, so a box without the built HIP extension fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VQA_HIP_LIBRARY: another build of the same ABI (the section-timing build of tools/section_profile.py)
LIB_PATH = os.environ.get("VQA_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libvqahip.so")

OK, EINVAL, ENOMEM, EHIP, EUNSUPPORTED, ENODEVICE = 0, -1, -2, -3, -4, -5

# every symbol include/vqa_hip.h declares
EXPORTS = (
    "vqa_version", "vqa_device_count", "vqa_last_error", "vqa_sync", "vqa_set_engine",
    "vqa_circuit_create", "vqa_circuit_destroy", "vqa_circuit_info", "vqa_circuit_engine",
    "vqa_observable_create", "vqa_observable_destroy",
    "vqa_eval_batch", "vqa_eval_batch_dev", "vqa_psr_full", "vqa_psr_full_dev",
    "vqa_statevector", "vqa_sample_counts", "vqa_sample_counts_dev", "vqa_dead_parameters", "vqa_psr_params_dev",
    "vqa_psr_full_noalloc_dev", "vqa_mse_stats_workspace", "vqa_mse_stats_dev", "vqa_mse_stats_shard_dev", "vqa_irandom_stats_dev", "vqa_class_grad_dev", "vqa_class_stats_dev", "vqa_sample_counts_shard_dev",
    "vqa_state_pass_counters", "vqa_light_cone_groups",
)


class VqaError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libvqahip error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C vqa_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback"
        )
    L = C.CDLL(LIB_PATH)
    vp, dp, ip, up = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    L.vqa_version.restype = C.c_int
    L.vqa_device_count.restype = C.c_int
    L.vqa_last_error.restype = C.c_char_p
    L.vqa_sync.argtypes = [C.c_int]
    L.vqa_set_engine.argtypes = [C.c_int]
    L.vqa_circuit_create.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.vqa_circuit_destroy.argtypes = [vp]
    L.vqa_circuit_info.argtypes = [vp] + [C.POINTER(C.c_int)] * 4
    L.vqa_circuit_engine.argtypes = [vp]
    L.vqa_circuit_engine.restype = C.c_int
    L.vqa_observable_create.argtypes = [C.c_int, dp, up, up, C.c_int, C.POINTER(vp)]
    L.vqa_observable_destroy.argtypes = [vp]
    L.vqa_eval_batch.argtypes = [vp, C.c_int, dp, dp, C.c_int, ip, ip, dp, C.c_int64,
                                 vp, C.c_int, C.c_int, dp, dp]
    L.vqa_eval_batch_dev.argtypes = [vp, C.c_int, vp, dp, dp, C.c_int, ip, ip, dp, C.c_int64,
                                     vp, C.c_int, C.c_int, dp, dp]
    L.vqa_psr_full.argtypes = [vp, C.c_int, dp, dp, C.c_int, ip, C.c_int, C.c_double,
                               vp, C.c_int, C.c_int, dp, dp]
    L.vqa_psr_full_dev.argtypes = [vp, C.c_int, vp, dp, dp, C.c_int, ip, C.c_int, C.c_double,
                                   vp, C.c_int, C.c_int, dp, dp]
    L.vqa_psr_full_noalloc_dev.argtypes = L.vqa_psr_full_dev.argtypes
    L.vqa_mse_stats_workspace.argtypes = [C.c_int64, C.c_int]
    L.vqa_mse_stats_dev.argtypes = [C.c_int, vp, dp, dp, C.c_int64, C.c_int, dp, C.c_double, dp, dp, C.c_double,
                                    C.c_int, C.c_uint64, C.c_int, C.c_double, C.c_int, dp, dp, dp, dp]
    L.vqa_mse_stats_shard_dev.argtypes = L.vqa_mse_stats_dev.argtypes + [C.c_int64]
    L.vqa_class_grad_dev.argtypes = [C.c_int, vp, ip, ip, C.c_int64, C.c_int, ip, dp, dp]
    L.vqa_class_stats_dev.argtypes = [C.c_int, vp, ip, ip, C.c_int64, C.c_int, ip, C.c_int, C.c_int, dp, dp]
    L.vqa_irandom_stats_dev.argtypes = [C.c_int, vp, dp, C.c_int, ip, dp, ip, ip, dp, C.c_int, C.c_int, C.c_int,
                                        C.c_double, C.c_double, C.c_uint64, dp]
    L.vqa_statevector.argtypes = [vp, C.c_int, dp, dp, C.c_int, C.c_double, dp]
    L.vqa_psr_params_dev.argtypes = [vp, C.c_int, vp, dp, dp, C.c_int, ip, C.c_int, C.c_double, vp,
                                     vp, C.c_int, C.c_int, dp, dp]
    L.vqa_dead_parameters.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.vqa_sample_counts.argtypes = [C.c_int, dp, C.c_int, ip, C.c_int64, C.c_uint64, C.c_uint64, ip]
    L.vqa_sample_counts_dev.argtypes = [C.c_int, vp, dp, C.c_int, ip, C.c_int64, C.c_uint64, C.c_uint64, ip]
    L.vqa_sample_counts_shard_dev.argtypes = L.vqa_sample_counts_dev.argtypes + [C.c_int64]
    L.vqa_state_pass_counters.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]
    L.vqa_light_cone_groups.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int64, vp, C.c_int]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ("vqa_last_error", "vqa_mse_stats_workspace"):
            fn.restype = C.c_int
    L.vqa_mse_stats_workspace.restype = C.c_int64
    _lib = L
    return L


def check(rc: int) -> int:
    if rc < 0:
        raise VqaError(rc, lib().vqa_last_error().decode("utf-8", "replace"))
    return rc


def nptr(a):
    """void* of a C-contiguous numpy array (or None)."""
    return None if a is None else C.c_void_p(a.ctypes.data)
