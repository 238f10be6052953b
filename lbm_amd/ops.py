"""The reference's operator functions on device tensors.

Each function keeps the name and argument meaning of the reference free function it replaces
(src/init.h:13-18, src/pull-push.h:73-140, src/io.h:25 / io.cpp:136-208) and forwards to the C ABI
(include/lbm_hip.h).  Populations are torch tensors of shape [9, NY, NX] (reference Dst_t with
COALESCING=true, global.h:112, main.cpp:37-38), float32 or float64, contiguous, on a HIP device.
Scalars that the reference passes as rank-0 device Views (DUI/DFL, global.h:121-147) are plain
Python numbers here.  All calls are asynchronous on the tensor's current torch stream unless noted.

`set_layout("aos")` switches every function to the reference's other layout, [NY, NX, 9]
(COALESCING=false, global.h:40, main.cpp:34-35); the switch is process-wide, like the macro.  A
`Context` carries layout, arithmetic and launch shape with an object instead (`with ctx: ...`).

There is no CPU path: a tensor that is not on a GPU is an error.
"""
from __future__ import annotations

import contextlib
import ctypes as C

import torch

from . import capi

Q = 9
SOA, AOS = "soa", "aos"
_LAYOUT_CODES = {SOA: 0, AOS: 1}
_layout = SOA


def set_layout(layout: str) -> None:
    """Layout of every population tensor from now on: "soa" = [9, NY, NX], "aos" = [NY, NX, 9]."""
    global _layout
    if layout not in _LAYOUT_CODES:
        raise ValueError(f"layout must be 'soa' or 'aos', got {layout!r}")
    capi.call("lbm_set_layout", _LAYOUT_CODES[layout])
    _layout = layout


def get_layout() -> str:
    """Layout in force for the calling thread: its current Context's, else the process-wide one."""
    lib = capi.load()
    cur = lib.lbm_ctx_get_current()
    code = lib.lbm_ctx_get_layout(cur) if cur else lib.lbm_get_layout()
    return AOS if code == _LAYOUT_CODES[AOS] else SOA


@contextlib.contextmanager
def layout(name: str):
    """`with ops.layout("aos"): ...` -- switch the process-wide layout for a block."""
    before = _layout
    set_layout(name)
    try:
        yield
    finally:
        set_layout(before)


def set_unroll_loops(unrolled: bool) -> None:
    """The reference's UNROLL_LOOPS switch (macros.h:6): False selects the arithmetic of
    pull_periodic_for / push_periodic_for (pull-push.cpp:77-120, :190-243) in the periodic sweeps,
    which differs from the default in the last bits.  Process-wide."""
    capi.call("lbm_set_unroll_loops", 1 if unrolled else 0)


def get_unroll_loops() -> bool:
    return bool(capi.load().lbm_get_unroll_loops())


ARITHMETICS = {"unrolled": 0, "rolled": 1, "contracted": 2}


def set_arithmetic(name: str) -> None:
    """Process-wide spelling of the relaxation: "contracted" (the default: explicit fused
    multiply-adds where the reference's own build flags allow a compiler to fuse), "unrolled" =
    "strict" (pull-push.cpp:25-72 with every operation rounded where the text puts it) or "rolled"
    (the reference's UNROLL_LOOPS=false functions)."""
    capi.call("lbm_set_arithmetic", ARITHMETICS["unrolled" if name == "strict" else name])


def get_arithmetic() -> str:
    """Arithmetic in force for the calling thread: its current Context's, else the process-wide one."""
    lib = capi.load()
    cur = lib.lbm_ctx_get_current()
    v = int(lib.lbm_ctx_get_arithmetic(cur) if cur else lib.lbm_get_arithmetic())
    return [k for k, x in ARITHMETICS.items() if x == v][0]


class Context:
    """Settings that travel with an object instead of with the process (lbm_ctx of the C ABI):
    layout, arithmetic and launch shape.  Unspecified settings are copied from those in force where
    the context is made.  `with ctx:` makes it current for the calling thread for the block, so that
    every operator called inside runs under its settings whatever the process-wide ones are; blocks
    nest."""

    def __init__(self, layout=None, arithmetic=None, threads_x=None, waves_per_cu=None):
        h = C.c_void_p()
        capi.call("lbm_ctx_create", C.byref(h))
        self._h = h
        self._before = []
        if layout is not None:
            if layout not in _LAYOUT_CODES:
                raise ValueError(f"layout must be 'soa' or 'aos', got {layout!r}")
            capi.call("lbm_ctx_set_layout", h, _LAYOUT_CODES[layout])
        if arithmetic is not None:
            capi.call("lbm_ctx_set_arithmetic", h, ARITHMETICS["unrolled" if arithmetic == "strict" else arithmetic])
        if threads_x is not None or waves_per_cu is not None:
            capi.call("lbm_ctx_set_launch_shape", h, int(threads_x or 0), int(waves_per_cu or 0))

    @property
    def layout(self) -> str:
        return AOS if capi.load().lbm_ctx_get_layout(self._h) == _LAYOUT_CODES[AOS] else SOA

    @property
    def arithmetic(self) -> str:
        v = int(capi.load().lbm_ctx_get_arithmetic(self._h))
        return [k for k, x in ARITHMETICS.items() if x == v][0]

    def __enter__(self):
        self._before.append(capi.load().lbm_ctx_get_current())
        capi.call("lbm_ctx_make_current", self._h)
        return self

    def __exit__(self, *exc):
        capi.call("lbm_ctx_make_current", C.c_void_p(self._before.pop()))
        return False

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and capi is not None:
            try:
                lib = capi.load()
                if lib.lbm_ctx_get_current() == h.value:
                    lib.lbm_ctx_make_current(None)
                lib.lbm_ctx_destroy(h)
            except Exception:  # interpreter shutdown
                pass


@contextlib.contextmanager
def arithmetic(name: str):
    """`with ops.arithmetic("contracted"): ...` -- switch the process-wide arithmetic for a block."""
    before = get_arithmetic()
    set_arithmetic(name)
    try:
        yield
    finally:
        set_arithmetic(before)


@contextlib.contextmanager
def unroll_loops(unrolled: bool):
    """`with ops.unroll_loops(False): ...` -- switch the process-wide arithmetic for a block."""
    before = get_unroll_loops()
    set_unroll_loops(unrolled)
    try:
        yield
    finally:
        set_unroll_loops(before)


_TUNING_DEFAULTS = {"fused_narrow": 1, "fused_group": -1, "fused_bulk": 1, "fused_max_steps": 0, "fused_min_nodes": 0,
                    "fused_lane_nodes": 0}


@contextlib.contextmanager
def tuning(name: str, value: int):
    """`with ops.tuning("fused_group", 1): ...` -- one of the library's tuning / test switches
    (lbm_set_tuning) for a block; the built-in value afterwards."""
    if name not in _TUNING_DEFAULTS:
        raise ValueError(f"unknown tuning switch {name!r}")
    capi.call("lbm_set_tuning", name.encode(), int(value))
    try:
        yield
    finally:
        capi.call("lbm_set_tuning", name.encode(), _TUNING_DEFAULTS[name])


def population_shape(nx: int, ny: int):
    """Shape of a population tensor of an nx x ny lattice in the current layout."""
    return (ny, nx, Q) if get_layout() == AOS else (Q, ny, nx)


def _prec(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return capi.F32
    if t.dtype == torch.float64:
        return capi.F64
    raise TypeError(f"populations must be float32 or float64, got {t.dtype}")


def _check_f(f: torch.Tensor, name: str = "f"):
    if not isinstance(f, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not f.is_cuda:
        raise RuntimeError(
            f"{name} lives on {f.device}: the lattice kernels are HIP-only (gfx950); there is no CPU path")
    if get_layout() == AOS:
        if f.dim() != 3 or f.shape[2] != Q or not f.is_contiguous():
            raise ValueError(f"{name} must be a contiguous [NY, NX, 9] tensor (layout 'aos'), got {tuple(f.shape)}")
        return int(f.shape[1]), int(f.shape[0])
    if f.dim() != 3 or f.shape[0] != Q or not f.is_contiguous():
        raise ValueError(f"{name} must be a contiguous [9, NY, NX] tensor, got {tuple(f.shape)}")
    return int(f.shape[2]), int(f.shape[1])


def _check_pair(f, buf, nx=None, ny=None):
    fx, fy = _check_f(f, "f")
    bx, by = _check_f(buf, "buf")
    if (fx, fy) != (bx, by) or f.dtype != buf.dtype or f.device != buf.device:
        raise ValueError("f and buf must have the same shape, dtype and device")
    if nx is not None and int(nx) != fx or ny is not None and int(ny) != fy:
        raise ValueError(f"nx,ny = {nx},{ny} do not match the tensor shape {fx},{fy}")
    return fx, fy


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _run(name: str, t: torch.Tensor, *args):
    with torch.cuda.device(t.device):
        capi.call(name, *args)


# ---------------------------------------------------------------- initialisation (init.h) ----
def init_shearwave(f, rho_init, u, y_below=0, y_glob=None, halo=0):
    """init_shearwave (init.h:13, init.cpp:5-34).  halo = the reference's USE_MPI row offset."""
    nx, ny = _check_f(f)
    y_glob = ny if y_glob is None else y_glob
    _run("lbm_init_shearwave", f, f.data_ptr(), _prec(f), nx, ny, float(rho_init), float(u),
         int(y_below), int(y_glob), int(halo), _stream(f))


def init_rest(f, rho_init):
    """init_rest (init.h:18, init.cpp:36-49)."""
    nx, ny = _check_f(f)
    _run("lbm_init_rest", f, f.data_ptr(), _prec(f), nx, ny, float(rho_init), _stream(f))


# ---------------------------------------------------------------- sweeps (pull-push.h) -------
def pull_periodic_no_mpi(f, buf, nx, ny, om):
    """pull_periodic_no_mpi (pull-push.h:84)."""
    nx, ny = _check_pair(f, buf, nx, ny)
    _run("lbm_pull_periodic_no_mpi", f, f.data_ptr(), buf.data_ptr(), _prec(f), nx, ny, float(om), _stream(f))


def push_periodic_no_mpi(f, buf, nx, ny, om):
    """push_periodic_no_mpi (pull-push.h:73)."""
    nx, ny = _check_pair(f, buf, nx, ny)
    _run("lbm_push_periodic_no_mpi", f, f.data_ptr(), buf.data_ptr(), _prec(f), nx, ny, float(om), _stream(f))


def pull_periodic_rows(f, buf, om, y_lo, y_hi, wrap_y):
    """Rows [y_lo,y_hi), x periodic; the slab form of pull_periodic_inner's y range (pull-push.h:95)."""
    nx, ny = _check_pair(f, buf)
    _run("lbm_pull_periodic_rows", f, f.data_ptr(), buf.data_ptr(), _prec(f), nx, ny, float(om),
         int(y_lo), int(y_hi), int(bool(wrap_y)), _stream(f))


def push_periodic_rows(f, buf, om, y_lo, y_hi, wrap_y):
    nx, ny = _check_pair(f, buf)
    _run("lbm_push_periodic_rows", f, f.data_ptr(), buf.data_ptr(), _prec(f), nx, ny, float(om),
         int(y_lo), int(y_hi), int(bool(wrap_y)), _stream(f))


def pull_periodic_inner(f, buf, om, LX, LY, HX, HY):
    """pull_periodic_inner (pull-push.h:95)."""
    nx, ny = _check_pair(f, buf)
    _run("lbm_pull_periodic_inner", f, f.data_ptr(), buf.data_ptr(), _prec(f), nx, ny, float(om),
         int(LX), int(LY), int(HX), int(HY), _stream(f))


def push_periodic_inner(f, buf, om, LX, LY, HX, HY):
    """push_periodic_inner (pull-push.h:107)."""
    nx, ny = _check_pair(f, buf)
    _run("lbm_push_periodic_inner", f, f.data_ptr(), buf.data_ptr(), _prec(f), nx, ny, float(om),
         int(LX), int(LY), int(HX), int(HY), _stream(f))


def pull_periodic_outer(f, buf, om, nx, ny):
    """pull_periodic_outer (pull-push.h:117)."""
    nx, ny = _check_pair(f, buf, nx, ny)
    _run("lbm_pull_periodic_outer", f, f.data_ptr(), buf.data_ptr(), _prec(f), nx, ny, float(om), _stream(f))


def push_periodic_outer(f, buf, om, nx, ny):
    """push_periodic_outer (pull-push.h:126)."""
    nx, ny = _check_pair(f, buf, nx, ny)
    _run("lbm_push_periodic_outer", f, f.data_ptr(), buf.data_ptr(), _prec(f), nx, ny, float(om), _stream(f))


def push_lid_driven(f, buf, nx, ny, om, rho_eq, u_lid):
    """push_lid_driven (pull-push.h:140)."""
    nx, ny = _check_pair(f, buf, nx, ny)
    _run("lbm_push_lid_driven", f, f.data_ptr(), buf.data_ptr(), _prec(f), nx, ny, float(om),
         float(rho_eq), float(u_lid), _stream(f))


def pull_lid_driven(f, buf, nx, ny, om, rho_eq, u_lid):
    """Pull form of the lid-driven cavity; the reference has no counterpart (main.cpp:72-74)."""
    nx, ny = _check_pair(f, buf, nx, ny)
    _run("lbm_pull_lid_driven", f, f.data_ptr(), buf.data_ptr(), _prec(f), nx, ny, float(om),
         float(rho_eq), float(u_lid), _stream(f))


def push_lid_rows(f, buf, om, rho_eq, u_lid, y_lo, y_hi, y_goff, ny_glob):
    nx, ny = _check_pair(f, buf)
    _run("lbm_push_lid_rows", f, f.data_ptr(), buf.data_ptr(), _prec(f), nx, ny, float(om),
         float(rho_eq), float(u_lid), int(y_lo), int(y_hi), int(y_goff), int(ny_glob), _stream(f))


def pull_lid_rows(f, buf, om, rho_eq, u_lid, y_lo, y_hi, y_goff, ny_glob):
    nx, ny = _check_pair(f, buf)
    _run("lbm_pull_lid_rows", f, f.data_ptr(), buf.data_ptr(), _prec(f), nx, ny, float(om),
         float(rho_eq), float(u_lid), int(y_lo), int(y_hi), int(y_goff), int(ny_glob), _stream(f))


# ---------------------------------------------------------------- several steps per call -----
def fused_max_steps(f) -> int:
    """Largest `nsteps` the fused sweeps accept for this array (1: rows are not whole 8-byte vectors)."""
    nx, _ = _check_f(f)
    return int(capi.load().lbm_fused_max_steps(_prec(f), nx))


def load_kernels(f) -> None:
    """Load the sweeps' code objects for f's precision and the settings in force onto f's device now
    instead of at the first launch (lbm_load_kernels)."""
    _check_f(f)
    with torch.cuda.device(f.device):
        capi.call("lbm_load_kernels", _prec(f))


def plan_steps(f, rows, remaining, lid: bool = False) -> int:
    """Steps the next launch of the time loop should take (1 .. fused_max_steps, <= remaining);
    lid: the launch is a lid-driven cavity sweep."""
    nx, _ = _check_f(f)
    return int(capi.load().lbm_plan_steps_scene(_prec(f), nx, int(rows), int(remaining), 1 if lid else 0))


def pull_periodic_fused(f, buf, om, y_lo, y_hi, wrap_y, nsteps):
    """`nsteps` iterations of the time loop (main.cpp:64-83) with pull_periodic for rows [y_lo, y_hi):
    f holds time t, buf receives time t + nsteps."""
    nx, ny = _check_pair(f, buf)
    _run("lbm_pull_periodic_fused", f, f.data_ptr(), buf.data_ptr(), _prec(f), nx, ny, float(om),
         int(y_lo), int(y_hi), int(bool(wrap_y)), int(nsteps), _stream(f))


def push_periodic_fused(f, buf, om, y_lo, y_hi, wrap_y, nsteps):
    nx, ny = _check_pair(f, buf)
    _run("lbm_push_periodic_fused", f, f.data_ptr(), buf.data_ptr(), _prec(f), nx, ny, float(om),
         int(y_lo), int(y_hi), int(bool(wrap_y)), int(nsteps), _stream(f))


def push_lid_fused(f, buf, om, rho_eq, u_lid, y_lo, y_hi, y_goff, ny_glob, nsteps):
    nx, ny = _check_pair(f, buf)
    _run("lbm_push_lid_fused", f, f.data_ptr(), buf.data_ptr(), _prec(f), nx, ny, float(om),
         float(rho_eq), float(u_lid), int(y_lo), int(y_hi), int(y_goff), int(ny_glob), int(nsteps), _stream(f))


def pull_lid_fused(f, buf, om, rho_eq, u_lid, y_lo, y_hi, y_goff, ny_glob, nsteps):
    nx, ny = _check_pair(f, buf)
    _run("lbm_pull_lid_fused", f, f.data_ptr(), buf.data_ptr(), _prec(f), nx, ny, float(om),
         float(rho_eq), float(u_lid), int(y_lo), int(y_hi), int(y_goff), int(ny_glob), int(nsteps), _stream(f))


# ---------------------------------------------------------------- output path (io.cpp) -------
def compute_velocities(vel, f, LX, LY, HX, HY):
    """compute_velocities (io.cpp:136-174).  vel: [2, HY-LY, NX] (ux plane, uy plane); the x range
    is always the full row (LX must be 0 and HX the row length)."""
    nx, ny = _check_f(f)
    if int(LX) != 0 or int(HX) != nx:
        raise ValueError("compute_velocities works on full rows: LX must be 0 and HX == NX")
    rows = int(HY) - int(LY)
    if tuple(vel.shape) != (2, rows, nx) or vel.dtype != f.dtype or not vel.is_contiguous() or vel.device != f.device:
        raise ValueError(f"vel must be a contiguous [2, {rows}, {nx}] tensor like f")
    _run("lbm_compute_velocities", f, f.data_ptr(), vel.data_ptr(), _prec(f), nx, ny, int(LY), int(HY), _stream(f))


def velocity_magnitudes(mag, f, LY, HY):
    """|u| per node as the -ov output forms it (io.cpp:236-238).  mag: [HY-LY, NX]."""
    nx, ny = _check_f(f)
    rows = int(HY) - int(LY)
    if tuple(mag.shape) != (rows, nx) or mag.dtype != f.dtype or not mag.is_contiguous() or mag.device != f.device:
        raise ValueError(f"mag must be a contiguous [{rows}, {nx}] tensor like f")
    _run("lbm_velocity_magnitudes", f, f.data_ptr(), mag.data_ptr(), _prec(f), nx, ny, int(LY), int(HY), _stream(f))


def max_velocity_dev(out, f, LY=0, HY=None):
    """MAX_VEL reduction of output() (io.cpp:184-208) into out[0] (1-element tensor like f, on the
    device); asynchronous."""
    nx, ny = _check_f(f)
    HY = ny if HY is None else int(HY)
    if out.numel() < 1 or out.dtype != f.dtype or out.device != f.device:
        raise ValueError("out must be a device tensor of f's dtype with at least one element")
    _run("lbm_max_velocity_dev", f, f.data_ptr(), out.data_ptr(), _prec(f), nx, ny, int(LY), HY, _stream(f))


def max_velocity(f, LY=0, HY=None) -> float:
    """MAX_VEL reduction delivered to the host (synchronises the stream)."""
    nx, ny = _check_f(f)
    HY = ny if HY is None else int(HY)
    out = C.c_double(0.0)
    _run("lbm_max_velocity", f, f.data_ptr(), _prec(f), nx, ny, int(LY), HY, _stream(f), C.byref(out))
    return out.value


def selftest_reciprocal(x: torch.Tensor) -> torch.Tensor:
    """1/x element-wise as the "contracted" arithmetic forms the reciprocal of the density
    (pull-push.cpp:27); the tests hold it to the correctly rounded quotient."""
    if not x.is_cuda or not x.is_contiguous():
        raise ValueError("x must be a contiguous device tensor")
    out = torch.empty_like(x)
    _run("lbm_selftest_reciprocal", x, x.data_ptr(), out.data_ptr(), _prec(x), x.numel(), _stream(x))
    return out
