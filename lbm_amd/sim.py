"""Single-device time loop and text outputs of the reference driver, on the HIP operators.

Mirrors run_simulation_single_node (reference src/main.cpp:31-89): allocate f/buf, initialise by
scene, then per step {optional output BEFORE the step, one fused sweep, handle swap}; and output()
(src/io.cpp:177-294) for the -omv / -ov / -ovf formats.
"""
from __future__ import annotations

import io as _io
from typing import Optional, TextIO

import torch

from . import ops

SHEAR_WAVE, LID_DRIVEN = "shear", "lid"
NONE, MAX_VEL, VEL_MAGS, VEL_FIELD = "none", "omv", "ov", "ovf"


def fmt(v: float) -> str:
    """C++ ostream default formatting of a float/double (precision 6, i.e. %g)."""
    return "%g" % v


class NaNEncountered(RuntimeError):
    """-ov output met a NaN (reference io.cpp:239-243: message on stderr, exit code 1)."""

    def __init__(self, x: int, y: int):
        super().__init__(f"NaN encountered at x={x} y={y}")
        self.x, self.y = x, y


class Simulation:
    """Two-grid, one-step D2Q9 BGK solver on one GPU.

    scene:  "shear" (periodic shear-wave decay, default) or "lid" (lid-driven cavity)
    scheme: "pull" or "push".  The reference's lid scene always pushes (main.cpp:72-74); here
            scheme="pull" with scene="lid" selects the pull form of the same boundary conditions.

    layout, arithmetic: "soa" / "aos" and "contracted" / "strict" / "rolled"; None takes what is in force
    where the object is made (ops.set_layout, ops.set_arithmetic, or an enclosing ops.Context).  The
    object keeps its settings in a context of its own (self.ctx) and runs every call under it, so
    simulations with different settings can be interleaved in one process and later changes of the
    process-wide settings do not reach it.
    """

    def __init__(self, nx: int, ny: int, scene: str = SHEAR_WAVE, scheme: str = "pull",
                 dtype: torch.dtype = torch.float32, omega: float = 0.5, rho: float = 1.0,
                 u: float = 0.1, device="cuda:0", layout: Optional[str] = None,
                 arithmetic: Optional[str] = None):
        if scene not in (SHEAR_WAVE, LID_DRIVEN):
            raise ValueError(f"unknown scene {scene!r}")
        if scheme not in ("pull", "push"):
            raise ValueError(f"unknown scheme {scheme!r}")
        self.nx, self.ny = int(nx), int(ny)
        self.scene, self.scheme = scene, scheme
        self.omega, self.rho, self.u = float(omega), float(rho), float(u)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("Simulation needs a HIP device; there is no CPU path")
        self.ctx = ops.Context(layout=layout, arithmetic=arithmetic)
        self.t = 0
        self.launches = 0   # sweep launches so far (a launch advances 1..8 steps)
        self.launch_steps = []   # steps taken by each of the most recent launches (bounded)
        self._vel: Optional[torch.Tensor] = None
        self._mag: Optional[torch.Tensor] = None
        with self.ctx:
            self.f = torch.empty(ops.population_shape(self.nx, self.ny), dtype=dtype, device=self.device)
            self.buf = torch.empty_like(self.f)  # needs no initialisation (main.cpp:206)
            ops.load_kernels(self.f)
            if scene == LID_DRIVEN:
                ops.init_rest(self.f, self.rho)
            else:
                ops.init_shearwave(self.f, self.rho, self.u, 0, self.ny)

    # -- n iterations of {one collide+stream sweep, handle swap} (main.cpp:71-84).  Lattices that
    #    stream from HBM take up to `max_fuse` iterations per launch (same result, one pass over
    #    memory); max_fuse=1 forces one launch per iteration.
    def step(self, n: int = 1, max_fuse: Optional[int] = None) -> None:
        with self.ctx:
            self._step(int(n), max_fuse)

    def plan(self, n: int) -> list:
        """Steps per launch that step(n) will take, in order."""
        out, n = [], int(n)
        with self.ctx:
            while n > 0:
                out.append(ops.plan_steps(self.f, self.ny, n, lid=self.scene == LID_DRIVEN))
                n -= out[-1]
        return out

    def _step(self, n: int, max_fuse: Optional[int]) -> None:
        while n > 0:
            k = ops.plan_steps(self.f, self.ny, n, lid=self.scene == LID_DRIVEN)
            if max_fuse is not None:
                k = max(1, min(k, int(max_fuse)))
            if k == 1:
                if self.scene == LID_DRIVEN:
                    if self.scheme == "push":
                        ops.push_lid_driven(self.f, self.buf, self.nx, self.ny, self.omega, self.rho, self.u)
                    else:
                        ops.pull_lid_driven(self.f, self.buf, self.nx, self.ny, self.omega, self.rho, self.u)
                elif self.scheme == "push":
                    ops.push_periodic_no_mpi(self.f, self.buf, self.nx, self.ny, self.omega)
                else:
                    ops.pull_periodic_no_mpi(self.f, self.buf, self.nx, self.ny, self.omega)
            elif self.scene == LID_DRIVEN:
                fused = ops.push_lid_fused if self.scheme == "push" else ops.pull_lid_fused
                fused(self.f, self.buf, self.omega, self.rho, self.u, 0, self.ny, 0, self.ny, k)
            else:
                fused = ops.push_periodic_fused if self.scheme == "push" else ops.pull_periodic_fused
                fused(self.f, self.buf, self.omega, 0, self.ny, True, k)
            self.f, self.buf = self.buf, self.f
            self.t += k
            self.launches += 1
            self.launch_steps.append(k)
            del self.launch_steps[:-4096]
            n -= k

    def max_velocity(self) -> float:
        with self.ctx:
            return ops.max_velocity(self.f)

    def velocities(self) -> torch.Tensor:
        """[2, NY, NX] device tensor: ux plane, uy plane."""
        if self._vel is None:
            self._vel = torch.empty((2, self.ny, self.nx), dtype=self.f.dtype, device=self.device)
        with self.ctx:
            ops.compute_velocities(self._vel, self.f, 0, 0, self.nx, self.ny)
        return self._vel

    def velocity_magnitudes(self) -> torch.Tensor:
        if self._mag is None:
            self._mag = torch.empty((self.ny, self.nx), dtype=self.f.dtype, device=self.device)
        with self.ctx:
            ops.velocity_magnitudes(self._mag, self.f, 0, self.ny)
        return self._mag

    # -- output() of io.cpp:177-294
    def output(self, kind: str, stream: TextIO) -> None:
        if kind == MAX_VEL:
            stream.write(fmt(self.max_velocity()) + ",")
            stream.flush()
        elif kind == VEL_MAGS:
            mag = self.velocity_magnitudes().cpu().numpy()
            _write_block(stream, mag, check_nan=True)
        elif kind == VEL_FIELD:
            vel = self.velocities().cpu().numpy()
            _write_block(stream, vel[0], check_nan=False)
            _write_block(stream, vel[1], check_nan=False)
        elif kind != NONE:
            raise ValueError(f"unknown output kind {kind!r}")

    def run(self, steps: int, out_every: int = 0, kind: str = NONE, stream: Optional[TextIO] = None) -> str:
        """The reference's main loop: output before the step when t % out_every == 0."""
        own = stream is None
        stream = _io.StringIO() if own else stream
        t = 0
        while t < steps:
            if out_every > 0 and t % out_every == 0:
                self.output(kind, stream)
            # up to the next output (or the end) in one go
            nxt = min(steps, (t // out_every + 1) * out_every) if out_every > 0 else steps
            self.step(nxt - t)
            t = nxt
        return stream.getvalue() if own else ""


def _write_block(stream: TextIO, a, check_nan: bool) -> None:
    """NY lines (y=0 first) of NX comma-separated values, then a '#' line (io.cpp:234-254)."""
    import numpy as np

    if check_nan and np.isnan(a).any():
        # the reference prints rows until it meets the NaN; report the first one in its scan order
        ys, xs = np.nonzero(np.isnan(a))
        for y in range(int(ys[0])):
            stream.write(",".join(fmt(v) for v in a[y]) + "\n")
        x = int(xs[0])
        if x > 0:
            stream.write(",".join(fmt(v) for v in a[int(ys[0]), :x]) + ",")
        raise NaNEncountered(x, int(ys[0]))
    for row in a:
        stream.write(",".join(fmt(v) for v in row) + "\n")
    stream.write("#\n")
