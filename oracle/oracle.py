"""ctypes/numpy front end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Loads ``oracle/liblbm_oracle.so`` (built by ``oracle/Makefile`` from the C restatement of the
reference's D2Q9 BGK path) and exposes it on numpy arrays laid out ``[9][NY][NX]``.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this
module; the product package ``lbm_amd`` must not.

Function names follow the reference's operator names (init.h:13-18, pull-push.h:73-140, io.h:25).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# "strict": -ffp-contract=off, the canonical checker (bit-comparable with the HIP kernels).
# "fma":    -ffp-contract=fast, what GCC does for the reference's own -O3 -march=native build
#           (CMakeLists.txt:18); used to show that this compiler freedom stays inside the parity
#           budget and to compare with results the reference's authors recorded from such a build.
_LIB_PATHS = {"strict": os.path.join(_HERE, "liblbm_oracle.so"),
              "fma": os.path.join(_HERE, "liblbm_oracle_fma.so")}
_LIB_PATH = _LIB_PATHS["strict"]
_libs = {}
VARIANT = "strict"

_u64 = C.c_uint64
_i64 = C.c_int64
_ptr = C.c_void_p


def build(force: bool = False) -> str:
    """Compile the oracle library in place (gcc, a second or two)."""
    if force or not all(os.path.exists(p) for p in _LIB_PATHS.values()):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


class variant:
    """Context manager selecting the oracle build: ``with orc.variant("fma"): ...``."""

    def __init__(self, name: str):
        if name not in _LIB_PATHS:
            raise ValueError(name)
        self.name = name

    def __enter__(self):
        global VARIANT
        self.prev, VARIANT = VARIANT, self.name
        return self

    def __exit__(self, *exc):
        global VARIANT
        VARIANT = self.prev
        return False


def lib() -> C.CDLL:
    _lib = _libs.get(VARIANT)
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATHS[VARIANT])
        for sfx, flt in (("_f32", C.c_float), ("_f64", C.c_double)):
            def sig(name, res, *args):
                fn = getattr(_lib, name + sfx)
                fn.restype = res
                fn.argtypes = list(args)
            sig("orc_init_shearwave", None, _ptr, _u64, _u64, flt, flt, _u64, _u64)
            sig("orc_init_rest", None, _ptr, _u64, _u64, flt)
            sig("orc_pull_rows", None, _ptr, _ptr, _u64, _u64, _u64, _u64, C.c_int, flt)
            sig("orc_push_rows", None, _ptr, _ptr, _u64, _u64, _u64, _u64, C.c_int, flt)
            sig("orc_pull_inner", None, _ptr, _ptr, _u64, _u64, flt, _u64, _u64, _u64, _u64)
            sig("orc_push_inner", None, _ptr, _ptr, _u64, _u64, flt, _u64, _u64, _u64, _u64)
            sig("orc_pull_outer", None, _ptr, _ptr, _u64, _u64, flt)
            sig("orc_push_outer", None, _ptr, _ptr, _u64, _u64, flt)
            sig("orc_push_lid_rows", None, _ptr, _ptr, _u64, _u64, _u64, _u64, _i64, _u64,
                flt, flt, flt)
            sig("orc_pull_lid_rows", None, _ptr, _ptr, _u64, _u64, _u64, _u64, _i64, _u64,
                flt, flt, flt)
            sig("orc_velocities", None, _ptr, _ptr, _u64, _u64, _u64, _u64)
            sig("orc_max_velocity", flt, _ptr, _u64, _u64, _u64, _u64)
            sig("orc_velocity_magnitudes", None, _ptr, _ptr, _u64, _u64, _u64, _u64)
        _libs[VARIANT] = _lib
    return _lib


def set_rolled(rolled: bool) -> None:
    """Periodic sweeps follow pull_periodic_for / push_periodic_for (the reference's
    UNROLL_LOOPS=false build, pull-push.cpp:77-120, :190-243) instead of the unrolled spelling."""
    lib().orc_set_rolled(1 if rolled else 0)


def get_rolled() -> bool:
    return bool(lib().orc_get_rolled())


ARITHMETICS = {"unrolled": 0, "rolled": 1, "contracted": 2}


def set_arithmetic(name: str) -> None:
    """Spelling of the relaxation: "unrolled" (pull-push.cpp:25-72, the default), "rolled" (the
    *_for functions, periodic sweeps only) or "contracted" (collide_fma: one explicit
    fused-multiply-add contraction of the unrolled spelling, the checker of the HIP library's
    LBM_ARITH_CONTRACTED mode)."""
    lib().orc_set_arithmetic(ARITHMETICS[name])


def get_arithmetic() -> str:
    v = int(lib().orc_get_arithmetic())
    return [k for k, x in ARITHMETICS.items() if x == v][0]


class arithmetic:
    """``with orc.arithmetic("contracted"): ...``"""

    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        self.prev = get_arithmetic()
        set_arithmetic(self.name)
        return self

    def __exit__(self, *exc):
        set_arithmetic(self.prev)
        return False


def num_threads() -> int:
    """OpenMP threads the oracle's row loops use."""
    fn = lib().orc_num_threads
    fn.restype = C.c_int
    return int(fn())


def _sfx(a: np.ndarray) -> str:
    if a.dtype == np.float32:
        return "_f32"
    if a.dtype == np.float64:
        return "_f64"
    raise TypeError(f"populations must be float32 or float64, got {a.dtype}")


def _check(f: np.ndarray) -> tuple[int, int]:
    if f.ndim != 3 or f.shape[0] != 9 or not f.flags.c_contiguous:
        raise ValueError("populations must be a C-contiguous [9][NY][NX] array")
    return f.shape[2], f.shape[1]


def _p(a: np.ndarray) -> int:
    return a.ctypes.data


def _call(name: str, a: np.ndarray, *args):
    return getattr(lib(), name + _sfx(a))(*args)


def alloc(nx: int, ny: int, dtype) -> np.ndarray:
    return np.empty((9, ny, nx), dtype=dtype)


# ---- initialisation (init.h:13, init.h:18) -------------------------------------------------
def init_shearwave(f, rho_init, u, y_below=0, y_glob=None):
    nx, ny = _check(f)
    _call("orc_init_shearwave", f, _p(f), nx, ny, rho_init, u, y_below,
          ny if y_glob is None else y_glob)


def init_rest(f, rho_init):
    nx, ny = _check(f)
    _call("orc_init_rest", f, _p(f), nx, ny, rho_init)


# ---- sweeps (pull-push.h:73-140) ------------------------------------------------------------
def pull_periodic_no_mpi(f, buf, om):
    nx, ny = _check(f); _check(buf)
    _call("orc_pull_rows", f, _p(f), _p(buf), nx, ny, 0, ny, 1, om)


def push_periodic_no_mpi(f, buf, om):
    nx, ny = _check(f); _check(buf)
    _call("orc_push_rows", f, _p(f), _p(buf), nx, ny, 0, ny, 1, om)


def pull_rows(f, buf, om, y_lo, y_hi, wrap_y):
    nx, ny = _check(f); _check(buf)
    _call("orc_pull_rows", f, _p(f), _p(buf), nx, ny, y_lo, y_hi, int(wrap_y), om)


def push_rows(f, buf, om, y_lo, y_hi, wrap_y):
    nx, ny = _check(f); _check(buf)
    _call("orc_push_rows", f, _p(f), _p(buf), nx, ny, y_lo, y_hi, int(wrap_y), om)


def pull_periodic_inner(f, buf, om, LX, LY, HX, HY):
    nx, ny = _check(f); _check(buf)
    _call("orc_pull_inner", f, _p(f), _p(buf), nx, ny, om, LX, LY, HX, HY)


def push_periodic_inner(f, buf, om, LX, LY, HX, HY):
    nx, ny = _check(f); _check(buf)
    _call("orc_push_inner", f, _p(f), _p(buf), nx, ny, om, LX, LY, HX, HY)


def pull_periodic_outer(f, buf, om):
    nx, ny = _check(f); _check(buf)
    _call("orc_pull_outer", f, _p(f), _p(buf), nx, ny, om)


def push_periodic_outer(f, buf, om):
    nx, ny = _check(f); _check(buf)
    _call("orc_push_outer", f, _p(f), _p(buf), nx, ny, om)


def push_lid_driven(f, buf, om, rho_eq, u_lid):
    nx, ny = _check(f); _check(buf)
    _call("orc_push_lid_rows", f, _p(f), _p(buf), nx, ny, 0, ny, 0, ny, om, rho_eq, u_lid)


def pull_lid_driven(f, buf, om, rho_eq, u_lid):
    """No reference counterpart (the reference's lid scene always pushes, main.cpp:72-74)."""
    nx, ny = _check(f); _check(buf)
    _call("orc_pull_lid_rows", f, _p(f), _p(buf), nx, ny, 0, ny, 0, ny, om, rho_eq, u_lid)


def push_lid_rows(f, buf, om, rho_eq, u_lid, y_lo, y_hi, y_goff, ny_glob):
    nx, ny = _check(f); _check(buf)
    _call("orc_push_lid_rows", f, _p(f), _p(buf), nx, ny, y_lo, y_hi, y_goff, ny_glob, om,
          rho_eq, u_lid)


def pull_lid_rows(f, buf, om, rho_eq, u_lid, y_lo, y_hi, y_goff, ny_glob):
    nx, ny = _check(f); _check(buf)
    _call("orc_pull_lid_rows", f, _p(f), _p(buf), nx, ny, y_lo, y_hi, y_goff, ny_glob, om,
          rho_eq, u_lid)


# ---- output path (io.cpp:136-294) -----------------------------------------------------------
def compute_velocities(f, y_lo=0, y_hi=None) -> np.ndarray:
    """[2][rows][NX]: ux plane then uy plane (io.cpp:136-174)."""
    nx, ny = _check(f)
    y_hi = ny if y_hi is None else y_hi
    vel = np.empty((2, y_hi - y_lo, nx), dtype=f.dtype)
    _call("orc_velocities", f, _p(f), _p(vel), nx, ny, y_lo, y_hi)
    return vel


def max_velocity(f, y_lo=0, y_hi=None):
    """io.cpp:184-208; returns a numpy scalar of the population dtype."""
    nx, ny = _check(f)
    y_hi = ny if y_hi is None else y_hi
    return f.dtype.type(_call("orc_max_velocity", f, _p(f), nx, ny, y_lo, y_hi))


def velocity_magnitudes(f, y_lo=0, y_hi=None) -> np.ndarray:
    nx, ny = _check(f)
    y_hi = ny if y_hi is None else y_hi
    mag = np.empty((y_hi - y_lo, nx), dtype=f.dtype)
    _call("orc_velocity_magnitudes", f, _p(f), _p(mag), nx, ny, y_lo, y_hi)
    return mag


def fmt(v) -> str:
    """C++ ostream default float formatting (precision 6, %g) used by io.cpp:219,245,271."""
    return "%g" % float(v)


def run(scene: str, scheme: str, nx: int, ny: int, steps: int, omega: float, dtype,
        rho: float = 1.0, u: float = 0.1, out_every: int = 0):
    """The reference's single-device time loop (main.cpp:31-89): init, then per step
    {optional max-|u| sample BEFORE the step; sweep; swap}.  Returns (f, samples)."""
    dt = np.dtype(dtype).type
    f, buf = alloc(nx, ny, dtype), alloc(nx, ny, dtype)
    om, rho, u = dt(omega), dt(rho), dt(u)
    if scene == "lid":
        init_rest(f, rho)
    else:
        init_shearwave(f, rho, u)
    samples = []
    for t in range(steps):
        if out_every > 0 and t % out_every == 0:
            samples.append(max_velocity(f))
        if scene == "lid":
            (pull_lid_driven if scheme == "pull" else push_lid_driven)(f, buf, om, rho, u)
        else:
            (pull_periodic_no_mpi if scheme == "pull" else push_periodic_no_mpi)(f, buf, om)
        f, buf = buf, f
    return f, samples
