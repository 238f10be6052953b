"""ctypes binding of include/lbm_hip.h (liblbm_hip.so, built in-tree by lbm_amd/build.py).

The library is the product: if it is missing or fails to load, importing callers get an
ImportError -- nothing here falls back to a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liblbm_hip.so")

F32, F64 = 4, 8
OK = 0

_u64, _i64, _int, _uint, _dbl, _ptr = C.c_uint64, C.c_int64, C.c_int, C.c_uint, C.c_double, C.c_void_p

# name -> (restype, argtypes); mirrors include/lbm_hip.h declaration by declaration
SIGNATURES = {
    "lbm_status_string": (C.c_char_p, [_int]),
    "lbm_last_error": (C.c_char_p, []),
    "lbm_abi_version": (_int, []),
    "lbm_set_layout": (_int, [_int]),
    "lbm_get_layout": (_int, []),
    "lbm_set_unroll_loops": (_int, [_int]),
    "lbm_get_unroll_loops": (_int, []),
    "lbm_set_arithmetic": (_int, [_int]),
    "lbm_get_arithmetic": (_int, []),
    "lbm_set_launch_shape": (_int, [_uint, _uint]),
    "lbm_load_kernels": (_int, [_int]),
    "lbm_ctx_create": (_int, [C.POINTER(_ptr)]),
    "lbm_ctx_destroy": (_int, [_ptr]),
    "lbm_ctx_make_current": (_int, [_ptr]),
    "lbm_ctx_get_current": (_ptr, []),
    "lbm_ctx_set_layout": (_int, [_ptr, _int]),
    "lbm_ctx_set_arithmetic": (_int, [_ptr, _int]),
    "lbm_ctx_set_launch_shape": (_int, [_ptr, _uint, _uint]),
    "lbm_ctx_get_layout": (_int, [_ptr]),
    "lbm_ctx_get_arithmetic": (_int, [_ptr]),
    "lbm_set_tuning": (_int, [C.c_char_p, _int]),
    "lbm_init_shearwave": (_int, [_ptr, _int, _u64, _u64, _dbl, _dbl, _u64, _u64, _uint, _ptr]),
    "lbm_init_rest": (_int, [_ptr, _int, _u64, _u64, _dbl, _ptr]),
    "lbm_pull_periodic_no_mpi": (_int, [_ptr, _ptr, _int, _u64, _u64, _dbl, _ptr]),
    "lbm_push_periodic_no_mpi": (_int, [_ptr, _ptr, _int, _u64, _u64, _dbl, _ptr]),
    "lbm_pull_periodic_rows": (_int, [_ptr, _ptr, _int, _u64, _u64, _dbl, _u64, _u64, _int, _ptr]),
    "lbm_push_periodic_rows": (_int, [_ptr, _ptr, _int, _u64, _u64, _dbl, _u64, _u64, _int, _ptr]),
    "lbm_pull_periodic_inner": (_int, [_ptr, _ptr, _int, _u64, _u64, _dbl, _u64, _u64, _u64, _u64, _ptr]),
    "lbm_push_periodic_inner": (_int, [_ptr, _ptr, _int, _u64, _u64, _dbl, _u64, _u64, _u64, _u64, _ptr]),
    "lbm_pull_periodic_outer": (_int, [_ptr, _ptr, _int, _u64, _u64, _dbl, _ptr]),
    "lbm_push_periodic_outer": (_int, [_ptr, _ptr, _int, _u64, _u64, _dbl, _ptr]),
    "lbm_push_lid_driven": (_int, [_ptr, _ptr, _int, _u64, _u64, _dbl, _dbl, _dbl, _ptr]),
    "lbm_pull_lid_driven": (_int, [_ptr, _ptr, _int, _u64, _u64, _dbl, _dbl, _dbl, _ptr]),
    "lbm_push_lid_rows": (_int, [_ptr, _ptr, _int, _u64, _u64, _dbl, _dbl, _dbl, _u64, _u64, _i64, _u64, _ptr]),
    "lbm_pull_lid_rows": (_int, [_ptr, _ptr, _int, _u64, _u64, _dbl, _dbl, _dbl, _u64, _u64, _i64, _u64, _ptr]),
    "lbm_fused_max_steps": (_int, [_int, _u64]),
    "lbm_plan_steps": (_int, [_int, _u64, _u64, _u64]),
    "lbm_plan_steps_scene": (_int, [_int, _u64, _u64, _u64, _int]),
    "lbm_pull_periodic_fused": (_int, [_ptr, _ptr, _int, _u64, _u64, _dbl, _u64, _u64, _int, _int, _ptr]),
    "lbm_push_periodic_fused": (_int, [_ptr, _ptr, _int, _u64, _u64, _dbl, _u64, _u64, _int, _int, _ptr]),
    "lbm_push_lid_fused": (_int, [_ptr, _ptr, _int, _u64, _u64, _dbl, _dbl, _dbl, _u64, _u64, _i64, _u64, _int, _ptr]),
    "lbm_pull_lid_fused": (_int, [_ptr, _ptr, _int, _u64, _u64, _dbl, _dbl, _dbl, _u64, _u64, _i64, _u64, _int, _ptr]),
    "lbm_compute_velocities": (_int, [_ptr, _ptr, _int, _u64, _u64, _u64, _u64, _ptr]),
    "lbm_velocity_magnitudes": (_int, [_ptr, _ptr, _int, _u64, _u64, _u64, _u64, _ptr]),
    "lbm_max_velocity_dev": (_int, [_ptr, _ptr, _int, _u64, _u64, _u64, _u64, _ptr]),
    "lbm_max_velocity_dev_static": (_int, [_ptr, _ptr, _int, _u64, _u64, _u64, _u64, _ptr]),
    "lbm_max_velocity": (_int, [_ptr, _int, _u64, _u64, _u64, _u64, _ptr, C.POINTER(_dbl)]),
    "lbm_selftest_reciprocal": (_int, [_ptr, _ptr, _int, _u64, _ptr]),
}

_lib = None


class LbmError(RuntimeError):
    """A C-ABI call returned a non-zero lbm_status."""

    def __init__(self, status: int, where: str, detail: str):
        super().__init__(f"{where}: status {status} ({detail})")
        self.status = status


def load() -> C.CDLL:
    """Load liblbm_hip.so and declare every prototype.  Raises ImportError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m lbm_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # e.g. libamdhip64 not found
        raise ImportError(f"cannot load {LIB_PATH}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    """Invoke a status-returning entry point; raise LbmError with the library's message on failure."""
    lib = load()
    status = getattr(lib, name)(*args)
    if status != OK:
        detail = lib.lbm_last_error().decode(errors="replace")
        kind = lib.lbm_status_string(status).decode()
        raise LbmError(status, name, f"{kind}: {detail}")
