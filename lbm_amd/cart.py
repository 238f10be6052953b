"""2-D Cartesian decomposition of the lattice over ranks (torch.distributed): the periodic shear
wave as the reference decomposes it, and the lid-driven cavity, which the reference does not.

The reference's MPI path (src/main.cpp:92-401): ranks form a periodic dims[0] x dims[1] grid, every
rank owns an nx x ny block of the lattice inside a local array with a halo of one node on every
side -- [9][ny+2][nx+2], owned nodes at 1..ny, 1..nx -- and one time step is

    exchange halos  ||  pull_periodic_inner on the nodes that touch no halo (main.cpp:321)
    pull_periodic_outer on the ring of owned nodes next to the halo (main.cpp:388), swap.

What differs from the reference, on purpose:
  * the exchange is device to device (RCCL send/recv when the backend is "nccl"), not staged
    through host mirrors (main.cpp:275-283, :330-338);
  * two phases instead of eight messages: columns first, then rows *including* the halo columns,
    so the four corner values (main.cpp:262-273, :363-374) ride along with the rows -- four
    messages per step and rank, each to a direct neighbour;
  * a halo column gets the populations that move in +x / -x and a halo row those that move in
    +y / -y, which is all the pull sweep reads there (the reference sends the same three per side);
  * blocks are initialised from the global row index; the reference's own offset
    (main.cpp:144) is wrong for ranks that do not sit in the bottom row of the grid.

Both streaming schemes of the periodic scene run here (the reference's lid-driven cavity refuses to
run under MPI, main.cpp:317-319), one time step per exchange:

  pull   exchange halos of f || pull_periodic_inner, then pull_periodic_outer, swap (as above);
  push   push_periodic_inner and push_periodic_outer scatter into buf including its halo ring; what
         landed in the ring has left the block and is *returned* to the neighbours: the halo column
         x = 0 (the populations that move in -x) goes to the left neighbour's last owned column,
         x = nx+1 to the right neighbour's first, over the full height so that the corner
         travellers ride along; then the halo rows to the lower / upper neighbours' edge rows.  The
         return of step t overlaps push_periodic_inner of step t+1, which reads no edge node.
         The reference's own push under MPI packs the edge populations of the pull scheme
         (main.cpp:238-261) and is not a correct exchange for it; this one is bit-identical to the
         single-device push run.

Blocks of several steps per exchange (`halo` > 1, the default wherever the multi-step sweeps take
the block): the local array carries a frame of G nodes on every side, [9][ny+2G][nx+2G] (one more
column on the right where that makes the row length even), holding plain copies of the
neighbours' outermost owned nodes, all nine populations.  One launch of the multi-step periodic
sweep advances rows [G, G+ny) of that array k <= G steps as if it were a lattice nx+2G wide that
wraps in x: what the wrap gets wrong travels one column per step and stays inside the frame, the
owned nodes come out exactly as on a single device.  Then the frame is refilled -- columns over
the owned rows first, then rows over the full width, so the corners ride along: four messages per
block of k steps, pull and push alike.  The exchange follows the sweep on the same stream (the
column phase needs every owned row, so there is no interior to overlap it with).

The lid-driven cavity (scene "lid") runs on the same frames with the multi-step cavity sweeps:
blocks carry a frame only towards neighbours, a block at a wall ends at its array's edge, where
the sweeps apply the wall (see CartSimulation).

For one node of GPUs the y-slab path (lbm_amd.slab, and `main -gpus N`) stays the fast
decomposition: it keeps rows whole and overlaps its exchange with the interior sweep.

Kernels are reached through a small backend object so that the schedule can be exercised on CPU
ranks in tests; the default backend is the HIP operator layer and refuses CPU tensors.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .slab import _edge_stream, ranks_share_device

# populations that move in +x, -x, +y, -y (D2Q9 numbering of global.h:86-93)
RIGHT_MOVERS = (1, 5, 8)
LEFT_MOVERS = (3, 6, 7)
UP_MOVERS = (2, 5, 6)
DOWN_MOVERS = (4, 7, 8)


def dims_create(world: int) -> Tuple[int, int]:
    """The most nearly square dims[0] >= dims[1] with dims[0] * dims[1] == world (what
    MPI_Dims_create(size, 2, dims) returns, main.cpp:103)."""
    if world < 1:
        raise ValueError("world size must be positive")
    d1 = int(world ** 0.5)
    while world % d1:
        d1 -= 1
    return world // d1, d1


def block_extent(n_glob: int, parts: int, coord: int) -> Tuple[int, int]:
    """(nodes owned, global index of the first owned node) along one axis: remainder nodes go to
    the lowest coordinates (main.cpp:131-141)."""
    base, rem = divmod(n_glob, parts)
    return base + (1 if coord < rem else 0), coord * base + min(coord, rem)


class HipCartBackend:
    """Block kernels on the HIP operator layer (lbm_amd.ops)."""

    def __init__(self):
        from . import ops
        self.ops = ops

    def init_shearwave(self, f, rho, u, y_below, y_glob, halo=1):
        self.ops.init_shearwave(f, rho, u, y_below, y_glob, halo=halo)

    def init_rest(self, f, rho):
        self.ops.init_rest(f, rho)

    def plan_steps(self, f, rows, remaining, lid=False):
        """Steps one launch should take on rows of f's length (1: one-step sweeps only)."""
        return self.ops.plan_steps(f, rows, remaining, lid=lid)

    def periodic_fused(self, scheme, f, buf, om, y_lo, y_hi, k):
        getattr(self.ops, f"{scheme}_periodic_fused")(f, buf, om, y_lo, y_hi, False, k)

    def lid_fused(self, scheme, f, buf, om, rho, u, y_lo, y_hi, y_goff, ny_glob, k):
        getattr(self.ops, f"{scheme}_lid_fused")(f, buf, om, rho, u, y_lo, y_hi, y_goff, ny_glob, k)

    def pull_inner(self, f, buf, om, LX, LY, HX, HY):
        self.ops.pull_periodic_inner(f, buf, om, LX, LY, HX, HY)

    def pull_outer(self, f, buf, om):
        self.ops.pull_periodic_outer(f, buf, om, f.shape[2], f.shape[1])

    def push_inner(self, f, buf, om, LX, LY, HX, HY):
        self.ops.push_periodic_inner(f, buf, om, LX, LY, HX, HY)

    def push_outer(self, f, buf, om):
        self.ops.push_periodic_outer(f, buf, om, f.shape[2], f.shape[1])

    def compute_velocities(self, vel, f, y_lo, y_hi):
        self.ops.compute_velocities(vel, f, 0, y_lo, f.shape[2], y_hi)


class CartSimulation:
    """Shear-wave decay (periodic) or the lid-driven cavity on a dims[0] x dims[1] grid of ranks, pull
    or push scheme.

    halo: frame depth = steps per exchange.  None (default) asks the library, 1 is the reference's
    scheme (inner sweep || exchange, outer ring; periodic scene only), > 1 runs blocks of up to that
    many steps in one launch of the multi-step sweep.

    scene "lid": the cavity of pull-push.cpp:432-531 cut in both directions, which the reference
    refuses to run under MPI (main.cpp:317-319).  Blocks carry a frame only towards neighbours: a
    block at a wall starts (ends) at its array's first (last) column or row, which is where the
    cavity sweeps apply the wall; towards a neighbour the sweeps see a wall at the end of the frame
    that is not there, and what that gets wrong stays inside the frame like the x wrap of the
    periodic scene.  Always on frames (any depth >= 1), never with the halo ring."""

    def __init__(self, nx_glob: int, ny_glob: int, dtype: torch.dtype = torch.float32, omega: float = 0.5,
                 rho: float = 1.0, u: float = 0.1, device="cuda:0", backend=None,
                 dims: Optional[Sequence[int]] = None, group=None, overlap: bool = True, scheme: str = "pull",
                 halo: Optional[int] = None, p2p_to_self: bool = False, scene: str = "shear"):
        if scheme not in ("pull", "push"):
            raise ValueError(f"unknown scheme {scheme!r}")
        if scene not in ("shear", "lid"):
            raise ValueError(f"unknown scene {scene!r}")
        self.scheme, self.scene = scheme, scene
        self.periodic = scene == "shear"
        # p2p_to_self: a rank that is its own neighbour along an axis still routes its halos through
        # the process group's send/recv (to itself) instead of local copies -- exercises the real
        # transport, e.g. RCCL on a one-GPU machine
        self.p2p_to_self = bool(p2p_to_self)
        self._outflow_pending = False   # push: the halo ring of f holds what the last step pushed out of the block
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.backend = backend if backend is not None else HipCartBackend()
        if backend is None and not self.cuda:
            raise RuntimeError("CartSimulation needs a HIP device; there is no CPU path")
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
            self.stream_ordered = self.cuda and dist.get_backend(group) == "nccl"
        else:
            self.world, self.rank, self.stream_ordered = 1, 0, False
            if self.p2p_to_self:
                raise RuntimeError("p2p_to_self needs an initialised process group")
        self.dims = tuple(int(d) for d in dims) if dims is not None else dims_create(self.world)
        if len(self.dims) != 2 or self.dims[0] * self.dims[1] != self.world:
            raise ValueError(f"dims {self.dims} do not multiply to the world size {self.world}")
        # row-major rank <-> coordinates, as MPI_Cart_create numbers them (main.cpp:104-106)
        self.coords = (self.rank // self.dims[1], self.rank % self.dims[1])
        self.nx_glob, self.ny_glob = int(nx_glob), int(ny_glob)
        self.nx, self.x_before = block_extent(self.nx_glob, self.dims[0], self.coords[0])
        self.ny, self.y_below = block_extent(self.ny_glob, self.dims[1], self.coords[1])
        # the smallest block of the grid, so that every rank raises or none does
        if self.nx_glob // self.dims[0] < 2 or self.ny_glob // self.dims[1] < 2:
            raise ValueError(f"a {self.nx_glob} x {self.ny_glob} lattice on a {self.dims[0]} x {self.dims[1]} grid of "
                             "ranks leaves blocks of fewer than 2 x 2 nodes")
        self.omega = float(omega)
        self.rho, self.u = float(rho), float(u)
        self.t = 0
        self._vel: Optional[torch.Tensor] = None
        if not self.periodic:
            self._setup_cavity(dtype, halo)
            return
        # Frame depth G.  halo=None: what the library plans per launch for the smallest block of the
        # grid (every rank must arrive at the same answer), as long as that block has G owned rows
        # and columns to send; 1 = the reference's scheme, one step per exchange.
        nx_min, ny_min = self.nx_glob // self.dims[0], self.ny_glob // self.dims[1]
        if halo is None:
            halo = 1
            if hasattr(self.backend, "plan_steps") and hasattr(self.backend, "periodic_fused"):
                halo = self._planned_depth(dtype, nx_min, ny_min)
        self.G = G = int(halo)
        if G < 1:
            raise ValueError(f"halo depth must be positive (got {halo})")
        if G > 1 and (nx_min < G or ny_min < G):
            raise ValueError(f"halo depth {G} needs blocks of at least {G} x {G} nodes")
        self.deep = G > 1
        self.x0 = self.y0 = G       # first owned column / row of the local arrays
        # deep frames: rows of an even length whatever nx is (the fp32 multi-step sweeps take two
        # nodes per lane); the extra column sits beyond the right frame and is never exchanged
        self.W = self.nx + 2 * G + ((self.nx + 2 * G) % 2 if self.deep else 0)
        self.f = torch.zeros((9, self.ny + 2 * G, self.W), dtype=dtype, device=self.device)
        self.buf = torch.zeros_like(self.f)
        if self.deep:
            # the wave is x-invariant: initialising whole rows fills the frame columns correctly
            self.backend.init_shearwave(self.f, float(rho), float(u), self.y_below, self.ny_glob, halo=G)
            self.buf.copy_(self.f)      # defined values everywhere a sweep may read
        else:
            self.backend.init_shearwave(self.f, float(rho), float(u), self.y_below, self.ny_glob)
        self._idx = {name: torch.tensor(ch, dtype=torch.long, device=self.device)
                     for name, ch in (("R", RIGHT_MOVERS), ("L", LEFT_MOVERS), ("U", UP_MOVERS), ("D", DOWN_MOVERS))}
        # pull: owned height / full width; push: full height (corner travellers) / owned width
        col_len, row_len = (self.ny, self.nx + 2) if scheme == "pull" else (self.ny + 2, self.nx)
        if self.deep:
            self._col = [torch.empty((9, self.ny, G), dtype=dtype, device=self.device) for _ in range(2)]
            self._row = [torch.empty((9, G, self.W), dtype=dtype, device=self.device) for _ in range(2)]
        else:
            self._col = [torch.empty((3, col_len), dtype=dtype, device=self.device) for _ in range(2)]
            self._row = [torch.empty((3, row_len), dtype=dtype, device=self.device) for _ in range(2)]
        # a second stream only where the exchange is stream-ordered (RCCL, or copies within the
        # rank); under the rehearsal transport the exchange blocks the host anyway
        self.overlap = bool(overlap) and self.cuda and (self.world == 1 or self.stream_ordered)
        # the device's one communication stream, shared with the y-slab path: see slab._edge_stream
        # for why every object must not take a fresh stream
        self._comm = (_edge_stream(self.device, high_priority=not ranks_share_device(self.device, self.group))
                      if (self.overlap and not self.deep) else None)
        if self.deep:
            self.overlap = False
            self._exchange_deep()       # ghost rows of the initial state (and the frame columns again)

    def _planned_depth(self, dtype, nx_min: int, ny_min: int, lid: bool = False, framed_x: bool = True) -> int:
        """Steps per launch for the smallest block of the grid, 1 if the multi-step sweeps do not
        take it.  Every block of the grid is at most one node wider; the row length is made even.
        framed_x=False: the block has no neighbour along x (a cavity on a 1 x N grid of ranks), so
        its rows get neither frames nor the extra column and keep the cavity's own length."""
        depth = 1
        for _ in range(2):      # the row length depends on the depth: settle in two rounds
            w = nx_min + (2 * depth if framed_x else 0)
            probe = torch.empty((9, 1, w + (w % 2 if framed_x else 0)), dtype=dtype, device=self.device)
            planned = int(self.backend.plan_steps(probe, ny_min, 1 << 30, lid) if lid
                          else self.backend.plan_steps(probe, ny_min, 1 << 30))
            depth = max(1, min(planned, nx_min, ny_min))
        return depth

    # ---- lid-driven cavity ----------------------------------------------------------------------
    def _setup_cavity(self, dtype, halo: Optional[int]) -> None:
        b = self.backend
        if not all(hasattr(b, name) for name in ("lid_fused", "plan_steps", "init_rest")):
            raise RuntimeError("the backend has no multi-step cavity sweep")
        cx, cy = self.coords
        # neighbours; None = wall
        self.left = self._rank_at(cx - 1, cy) if cx > 0 else None
        self.right = self._rank_at(cx + 1, cy) if cx < self.dims[0] - 1 else None
        self.lower = self._rank_at(cx, cy - 1) if cy > 0 else None
        self.upper = self._rank_at(cx, cy + 1) if cy < self.dims[1] - 1 else None
        nx_min, ny_min = self.nx_glob // self.dims[0], self.ny_glob // self.dims[1]
        if halo is None:
            halo = self._planned_depth(dtype, nx_min, ny_min, lid=True, framed_x=self.dims[0] > 1)
        self.G = G = int(halo)
        if G < 1:
            raise ValueError(f"halo depth must be positive (got {halo})")
        if nx_min < G or ny_min < G:
            raise ValueError(f"halo depth {G} needs blocks of at least {G} x {G} nodes")
        self.deep = True
        fl, fr = (G if self.left is not None else 0), (G if self.right is not None else 0)
        fb, ft = (G if self.lower is not None else 0), (G if self.upper is not None else 0)
        # rows of an even length (the fp32 multi-step sweeps take two nodes per lane): one extra
        # column beyond a frame, never on a wall side -- the wall must stay the array's end
        pad_l = pad_r = 0
        if (self.nx + fl + fr) % 2:
            if self.right is not None:
                pad_r = 1
            elif self.left is not None:
                pad_l = 1
        self.x0, self.y0 = pad_l + fl, fb
        self.W = pad_l + fl + self.nx + fr + pad_r
        self.f = torch.zeros((9, fb + self.ny + ft, self.W), dtype=dtype, device=self.device)
        self.buf = torch.zeros_like(self.f)
        b.init_rest(self.f, self.rho)          # frames and ghost rows included: the fluid is at rest everywhere
        self.buf.copy_(self.f)
        self._col = [torch.empty((9, self.ny, G), dtype=dtype, device=self.device) for _ in range(2)]
        self._row = [torch.empty((9, G, self.W), dtype=dtype, device=self.device) for _ in range(2)]
        self.overlap = False
        self._comm = None

    def _swap_open(self, send_hi, send_lo, recv_lo, recv_hi, hi_rank: Optional[int], lo_rank: Optional[int]) -> bool:
        """_swap along an axis that does not wrap: a side whose rank is None is a wall, nothing goes
        there and nothing comes from there.  Between two ranks at most one message travels in each
        direction, so posting order cannot mix them up."""
        staged = self.cuda and not self.stream_ordered
        ops, landed = [], []
        if hi_rank is not None:
            ops.append(dist.P2POp(dist.isend, send_hi.cpu() if staged else send_hi, self._peer(hi_rank), self.group))
        if lo_rank is not None:
            ops.append(dist.P2POp(dist.isend, send_lo.cpu() if staged else send_lo, self._peer(lo_rank), self.group))
        for rank, into in ((lo_rank, recv_lo), (hi_rank, recv_hi)):
            if rank is not None:
                box = torch.empty(into.shape, dtype=into.dtype, device="cpu") if staged else into
                ops.append(dist.P2POp(dist.irecv, box, self._peer(rank), self.group))
                landed.append((into, box))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if staged:
            for into, box in landed:
                into.copy_(box)
        return bool(ops)

    def _exchange_cavity(self) -> None:
        """Refill the frames of f from the neighbours' owned nodes, all nine populations: columns
        over the owned rows, then rows over the full width (the corners ride along)."""
        f, nx, ny, G, x0, y0 = self.f, self.nx, self.ny, self.G, self.x0, self.y0
        rows = slice(y0, y0 + ny)
        to_right = f[:, rows, x0 + nx - G:x0 + nx].contiguous() if self.right is not None else None
        to_left = f[:, rows, x0:x0 + G].contiguous() if self.left is not None else None
        self._swap_open(to_right, to_left, self._col[0], self._col[1], self.right, self.left)
        if self.left is not None:
            f[:, rows, x0 - G:x0] = self._col[0]
        if self.right is not None:
            f[:, rows, x0 + nx:x0 + nx + G] = self._col[1]
        to_up = f[:, y0 + ny - G:y0 + ny, :].contiguous() if self.upper is not None else None
        to_down = f[:, y0:y0 + G, :].contiguous() if self.lower is not None else None
        self._swap_open(to_up, to_down, self._row[0], self._row[1], self.upper, self.lower)
        if self.lower is not None:
            f[:, y0 - G:y0, :] = self._row[0]
        if self.upper is not None:
            f[:, y0 + ny:y0 + ny + G, :] = self._row[1]

    # ---- neighbours ---------------------------------------------------------------------------
    def _rank_at(self, cx: int, cy: int) -> int:
        return (cx % self.dims[0]) * self.dims[1] + (cy % self.dims[1])

    def _peer(self, group_rank: int) -> int:
        return dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank

    def _swap(self, send_hi, send_lo, recv_lo, recv_hi, hi_rank: int, lo_rank: int) -> None:
        """send_hi goes to the neighbour at the higher coordinate and arrives in its recv_lo;
        send_lo goes to the lower neighbour and arrives in its recv_hi."""
        if hi_rank == self.rank and not self.p2p_to_self:   # one rank along this axis: the lattice wraps onto itself
            recv_lo.copy_(send_hi)
            recv_hi.copy_(send_lo)
            return
        # A transport that is not stream-ordered (gloo) gets host tensors: what it does with device
        # pointers is its own business, and a receive that lands in device memory behind the GPU's
        # caches was seen to leave the next kernel with the previous step's values now and then.
        staged = self.cuda and not self.stream_ordered
        if staged:
            out_hi, out_lo = send_hi.cpu(), send_lo.cpu()        # synchronous copies
            in_lo, in_hi = torch.empty_like(out_hi), torch.empty_like(out_lo)
        else:
            out_hi, out_lo, in_lo, in_hi = send_hi, send_lo, recv_lo, recv_hi
        # same order on every rank: with two ranks along an axis both messages go to one peer and
        # are matched in the order they were posted
        ops = [dist.P2POp(dist.isend, out_hi, self._peer(hi_rank), self.group),
               dist.P2POp(dist.isend, out_lo, self._peer(lo_rank), self.group),
               dist.P2POp(dist.irecv, in_lo, self._peer(lo_rank), self.group),
               dist.P2POp(dist.irecv, in_hi, self._peer(hi_rank), self.group)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if staged:
            recv_lo.copy_(in_lo)
            recv_hi.copy_(in_hi)

    def _exchange(self) -> None:
        f, nx, ny = self.f, self.nx, self.ny
        cx, cy = self.coords
        R, L, U, D = (self._idx[k] for k in "RLUD")
        # columns: what moves in +x leaves through the last owned column, what moves in -x through
        # the first (main.cpp:250-261 packs the same populations)
        to_right = f[R, 1:ny + 1, nx]
        to_left = f[L, 1:ny + 1, 1]
        self._swap(to_right, to_left, self._col[0], self._col[1], self._rank_at(cx + 1, cy), self._rank_at(cx - 1, cy))
        f[R, 1:ny + 1, 0] = self._col[0]
        f[L, 1:ny + 1, nx + 1] = self._col[1]
        # rows, halo columns included: the corner nodes' diagonal movers travel with them
        to_up = f[U, ny, :]
        to_down = f[D, 1, :]
        self._swap(to_up, to_down, self._row[0], self._row[1], self._rank_at(cx, cy + 1), self._rank_at(cx, cy - 1))
        f[U, 0, :] = self._row[0]
        f[D, ny + 1, :] = self._row[1]

    def _exchange_deep(self) -> None:
        """Refill the frame of f from the neighbours' owned nodes, all nine populations: columns
        over the owned rows, then rows over the full width (the corners ride along)."""
        f, nx, ny, G = self.f, self.nx, self.ny, self.G
        cx, cy = self.coords
        to_right = f[:, G:G + ny, nx:nx + G].contiguous()        # my last G owned columns
        to_left = f[:, G:G + ny, G:2 * G].contiguous()           # my first G owned columns
        self._swap(to_right, to_left, self._col[0], self._col[1], self._rank_at(cx + 1, cy), self._rank_at(cx - 1, cy))
        f[:, G:G + ny, 0:G] = self._col[0]
        f[:, G:G + ny, nx + G:nx + 2 * G] = self._col[1]
        to_up = f[:, ny:ny + G, :].contiguous()                  # my highest G owned rows
        to_down = f[:, G:2 * G, :].contiguous()                  # my lowest G owned rows
        self._swap(to_up, to_down, self._row[0], self._row[1], self._rank_at(cx, cy + 1), self._rank_at(cx, cy - 1))
        f[:, 0:G, :] = self._row[0]
        f[:, ny + G:ny + 2 * G, :] = self._row[1]

    def _block_deep(self, k: int) -> None:
        """k <= G time steps of the block in one launch, then the frame of the new state."""
        G = self.G
        if self.periodic:
            self.backend.periodic_fused(self.scheme, self.f, self.buf, self.omega, G, G + self.ny, k)
        else:
            self.backend.lid_fused(self.scheme, self.f, self.buf, self.omega, self.rho, self.u, self.y0,
                                   self.y0 + self.ny, self.y_below - self.y0, self.ny_glob, k)
        self.f, self.buf = self.buf, self.f
        if self.periodic:
            self._exchange_deep()
        else:
            self._exchange_cavity()
        self.t += k

    def _return_outflow(self) -> None:
        """Push scheme: what the last step scattered into the halo ring of f goes to the neighbours'
        edge nodes (the reverse of _exchange).  Columns over the full height first -- the entries in
        the halo rows are the corner travellers, which the row phase then forwards."""
        f, nx, ny = self.f, self.nx, self.ny
        cx, cy = self.coords
        R, L, U, D = (self._idx[k] for k in "RLUD")
        out_right = f[R, :, nx + 1]
        out_left = f[L, :, 0]
        self._swap(out_right, out_left, self._col[0], self._col[1], self._rank_at(cx + 1, cy), self._rank_at(cx - 1, cy))
        f[R, :, 1] = self._col[0]          # what my left neighbour pushed to the right arrives in my first column
        f[L, :, nx] = self._col[1]
        out_up = f[U, ny + 1, 1:nx + 1]
        out_down = f[D, 0, 1:nx + 1]
        self._swap(out_up, out_down, self._row[0], self._row[1], self._rank_at(cx, cy + 1), self._rank_at(cx, cy - 1))
        f[U, 1, 1:nx + 1] = self._row[0]
        f[D, ny, 1:nx + 1] = self._row[1]
        self._outflow_pending = False

    def _complete(self) -> None:
        """Make f the complete state of time t (push: collect the outflow of the last step)."""
        if self._outflow_pending:
            self._return_outflow()

    def _step_push(self) -> None:
        NX, NY = self.nx + 2, self.ny + 2
        if self.overlap and self._outflow_pending:
            main = torch.cuda.current_stream(self.device)
            self._comm.wait_stream(main)
            self.backend.push_inner(self.f, self.buf, self.omega, 2, 2, NX - 2, NY - 2)
            with torch.cuda.stream(self._comm):
                self._return_outflow()
            main.wait_stream(self._comm)
        else:
            self._complete()
            self.backend.push_inner(self.f, self.buf, self.omega, 2, 2, NX - 2, NY - 2)
        self.backend.push_outer(self.f, self.buf, self.omega)
        self.f, self.buf = self.buf, self.f
        self._outflow_pending = True
        self.t += 1

    # ---- one time step (main.cpp:217-398) ------------------------------------------------------
    def step(self, nsteps: int = 1) -> None:
        if self.deep:
            # the fewest blocks of at most G steps, as equal as possible
            left = int(nsteps)
            while left > 0:
                blocks = -(-left // self.G)
                k = -(-left // blocks)
                self._block_deep(k)
                left -= k
            return
        for _ in range(int(nsteps)):
            if self.scheme == "push":
                self._step_push()
                continue
            NX, NY = self.nx + 2, self.ny + 2
            if self.overlap:
                # the exchange waits for the previous step only, not for the inner sweep launched
                # next to it
                main = torch.cuda.current_stream(self.device)
                self._comm.wait_stream(main)
                self.backend.pull_inner(self.f, self.buf, self.omega, 2, 2, NX - 2, NY - 2)
                with torch.cuda.stream(self._comm):
                    self._exchange()
                main.wait_stream(self._comm)
            else:
                self._exchange()
                self.backend.pull_inner(self.f, self.buf, self.omega, 2, 2, NX - 2, NY - 2)
            self.backend.pull_outer(self.f, self.buf, self.omega)
            self.f, self.buf = self.buf, self.f
            self.t += 1

    # ---- outputs --------------------------------------------------------------------------------
    def velocities(self) -> torch.Tensor:
        """ux, uy of the owned nodes: [2, ny, nx]."""
        self._complete()
        if self._vel is None:
            self._vel = torch.empty((2, self.ny, self.f.shape[2]), dtype=self.f.dtype, device=self.device)
        self.backend.compute_velocities(self._vel, self.f, self.y0, self.ny + self.y0)
        return self._vel[:, :, self.x0:self.nx + self.x0]

    def max_velocity(self) -> float:
        """Global max |u| (io.cpp:184-216: float square root, NaN never wins, MPI_Reduce(MAX))."""
        v = self.velocities()
        speed = torch.sqrt((v[0] * v[0] + v[1] * v[1]).to(torch.float32))
        speed = torch.where(torch.isnan(speed), torch.full_like(speed, -1e30), speed)
        m = speed.max().reshape(1)
        if self.world > 1:
            if self.cuda and not self.stream_ordered:
                m = m.cpu()
            dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
        return float(m.item())

    def interior(self) -> torch.Tensor:
        """The owned nodes without the halo: [9, ny, nx] (a view)."""
        self._complete()
        return self.f[:, self.y0:self.ny + self.y0, self.x0:self.nx + self.x0]

    def gather_populations(self) -> Optional[torch.Tensor]:
        """Rank 0 gets the global [9, NY, NX] populations (for tests / small outputs)."""
        mine = self.interior().contiguous()
        if self.world == 1:
            return mine
        staged = self.cuda and not self.stream_ordered     # see _swap
        if self.rank != 0:
            dist.send(mine.cpu() if staged else mine, self._peer(0), group=self.group)
            return None
        out = torch.empty((9, self.ny_glob, self.nx_glob), dtype=mine.dtype, device=self.device)
        for r in range(self.world):
            cx, cy = r // self.dims[1], r % self.dims[1]
            nx, x0 = block_extent(self.nx_glob, self.dims[0], cx)
            ny, y0 = block_extent(self.ny_glob, self.dims[1], cy)
            part = mine
            if r != 0:
                part = torch.empty((9, ny, nx), dtype=mine.dtype, device="cpu" if staged else self.device)
                dist.recv(part, self._peer(r), group=self.group)
            out[:, y0:y0 + ny, x0:x0 + nx] = part.to(self.device)
        return out

    def synchronize(self) -> None:
        if self.cuda:
            torch.cuda.synchronize(self.device)
