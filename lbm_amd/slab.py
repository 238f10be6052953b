"""y-slab decomposition of the lattice over ranks (one process per GPU, torch.distributed).

Replaces the reference's 2-D Cartesian MPI halo exchange (src/main.cpp:92-401) for one node of
GPUs: the lattice is cut into contiguous y-slabs, every slab spans the full NX so the x wrap stays
inside the sweep kernel, and each rank keeps one ghost row below and above its rows in every
population plane -- local array [9][n+2][NX], interior rows 1..n.  In SoA storage a (channel, row)
pair is already contiguous, so the three populations that cross a slab edge are sent straight from
/ received straight into the population array: no pack or unpack kernels (the reference needs six,
main.cpp:238-273, :339-374).

Two schedules share this layout.  Lattices that stream from HBM (and whose rows are whole 16-byte
vectors) run the deep-halo schedule: G ghost rows per side (the steps one launch takes: 7 or 8
on large slabs) hold plain copies of the neighbour's
edge rows, a block of up to G time steps is advanced by one fused launch per region (see
csrc/lbm_fused.h: every population crosses HBM once per block, not once per step), and ghost rows
are refreshed once per block -- one message of nine G-row blocks per direction.  Everything else
runs the one-row schedule described next.  In both, the interior and the edge regions of a slab
are advanced on two streams so that the exchange hides behind the interior update.

Per step, two streams run side by side.  The edge stream updates rows 1 and n and then exchanges
halo rows with the neighbours (RCCL send/recv over xGMI when the backend is "nccl"); the interior
stream updates rows 2..n-1.  The interior sweep of step t+1 needs only the interior and edge
sweeps of step t, never the exchange, so it is never held up by communication; the edge sweeps of
step t+1 follow the exchange of step t on their own stream.  Pull sends the freshly computed edge
rows into the neighbours' ghost rows; push sends what was pushed into the ghost rows to the
neighbours' edge rows.

The multi-rank result is bit-identical to the single-device result: slabs are initialised from the
global row index (the reference's own offset is wrong for ranks > 0, main.cpp:141-144, and is not
reproduced).

Kernels are reached through a small backend object so that the exchange schedule can be exercised
on CPU ranks in tests; the default backend is the HIP operator layer and refuses CPU tensors.
"""
from __future__ import annotations

import os

from typing import Optional, Tuple

import torch
import torch.distributed as dist

SHEAR_WAVE, LID_DRIVEN = "shear", "lid"

# populations that leave a slab through its top edge (cy=+1) / bottom edge (cy=-1)
UP_CHANNELS = (2, 5, 6)
DOWN_CHANNELS = (4, 7, 8)


def slab_rows(ny_glob: int, world: int, rank: int) -> Tuple[int, int]:
    """(rows owned, global index of the first owned row): remainder rows go to the lowest ranks
    (as reference main.cpp:133-141 does)."""
    base, rem = divmod(ny_glob, world)
    n = base + (1 if rank < rem else 0)
    y_below = rank * base + min(rank, rem)
    return n, y_below


class HipBackend:
    """Slab kernels on the HIP operator layer (lbm_amd.ops)."""

    def __init__(self):
        from . import ops
        self.ops = ops

    def init_shearwave(self, f, rho, u, y_below, y_glob, halo=1):
        self.ops.init_shearwave(f, rho, u, y_below, y_glob, halo=halo)

    def plan_steps(self, f, rows, remaining, lid=False):
        return self.ops.plan_steps(f, rows, remaining, lid=lid)

    def periodic_fused(self, scheme, f, buf, om, y_lo, y_hi, k):
        fn = self.ops.pull_periodic_fused if scheme == "pull" else self.ops.push_periodic_fused
        fn(f, buf, om, y_lo, y_hi, False, k)

    def lid_fused(self, scheme, f, buf, om, rho, u, y_lo, y_hi, y_goff, ny_glob, k):
        fn = self.ops.pull_lid_fused if scheme == "pull" else self.ops.push_lid_fused
        fn(f, buf, om, rho, u, y_lo, y_hi, y_goff, ny_glob, k)

    def init_rest(self, f, rho):
        self.ops.init_rest(f, rho)

    def pull_periodic_rows(self, f, buf, om, y_lo, y_hi):
        self.ops.pull_periodic_rows(f, buf, om, y_lo, y_hi, False)

    def push_periodic_rows(self, f, buf, om, y_lo, y_hi):
        self.ops.push_periodic_rows(f, buf, om, y_lo, y_hi, False)

    def pull_lid_rows(self, f, buf, om, rho, u, y_lo, y_hi, y_goff, ny_glob):
        self.ops.pull_lid_rows(f, buf, om, rho, u, y_lo, y_hi, y_goff, ny_glob)

    def push_lid_rows(self, f, buf, om, rho, u, y_lo, y_hi, y_goff, ny_glob):
        self.ops.push_lid_rows(f, buf, om, rho, u, y_lo, y_hi, y_goff, ny_glob)

    def max_velocity_dev(self, out, f, y_lo, y_hi):
        self.ops.max_velocity_dev(out, f, y_lo, y_hi)

    def compute_velocities(self, vel, f, y_lo, y_hi):
        self.ops.compute_velocities(vel, f, 0, y_lo, f.shape[2], y_hi)


_EDGE_STREAMS = {}


def _edge_stream(device: torch.device, high_priority: bool = True) -> "torch.cuda.Stream":
    """The one edge/communication stream of a device, shared by every slab on it.

    Created once per process and device.  Each new high-priority stream makes the HIP runtime open
    one more high-priority hardware queue on first use, up to GPU_MAX_HW_QUEUES (4 by default), and
    on this platform the device was seen to stop draining exactly when a process's last permitted
    high-priority queue starts taking cross-stream event waits (fourth stream by default, third
    with the limit at 3, never with the limit at 2 or with default-priority streams): a 40-line
    torch-only program with the same stream and event pattern reproduces it, with local copies in
    place of RCCL and no kernel of this package.  The schedule's event graph is acyclic (every wait
    names an event recorded earlier in host order), so slab objects share one high-priority queue
    instead of taking a stream each; slabs that share it are ordered behind each other on it.

    high_priority=False: a stream of default priority (what ranks_share_device() asks for: several
    processes on one device each bring their own queues, and default-priority streams never showed
    the stall).  The first request for a device decides; later ones get the same stream.
    """
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    s = _EDGE_STREAMS.get(key)
    if s is None:
        s = _EDGE_STREAMS[key] = torch.cuda.Stream(device, priority=-1 if high_priority else 0)
    return s


def ranks_share_device(device: torch.device, group=None) -> bool:
    """True when another rank of the process group runs on the same physical device as this one
    (rehearsals of a multi-rank run on fewer GPUs than ranks; a production run has one process per
    GPU).  Such ranks take their edge stream at default priority: see _edge_stream."""
    if device.type != "cuda" or not (dist.is_available() and dist.is_initialized()):
        return False
    world = dist.get_world_size(group)
    if world < 2:
        return False
    import socket
    props = torch.cuda.get_device_properties(device)
    ident = (socket.gethostname(), getattr(props, "uuid", None) and str(props.uuid),
             getattr(props, "pci_domain_id", None), getattr(props, "pci_bus_id", None), getattr(props, "pci_device_id", None),
             None if getattr(props, "pci_bus_id", None) is not None else
             (os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "")), device.index))
    everyone = [None] * world
    dist.all_gather_object(everyone, ident, group=group)
    return everyone.count(ident) > 1


class SlabSimulation:
    """One rank's slab of a global NX x NY lattice.

    Requires an initialised torch.distributed process group unless world size is 1 (then the
    periodic neighbours are the rank itself and rows are copied locally).
    """

    def __init__(self, nx: int, ny_glob: int, scene: str = SHEAR_WAVE, scheme: str = "pull",
                 dtype: torch.dtype = torch.float32, omega: float = 0.5, rho: float = 1.0,
                 u: float = 0.1, device="cuda:0", backend=None, group=None, overlap: bool = True,
                 p2p_to_self: bool = False, halo: Optional[int] = None):
        if scene not in (SHEAR_WAVE, LID_DRIVEN):
            raise ValueError(f"unknown scene {scene!r}")
        if scheme not in ("pull", "push"):
            raise ValueError(f"unknown scheme {scheme!r}")
        self.group = group
        # p2p_to_self: with a single rank, still route the periodic halo rows through the process
        # group's send/recv (to this rank itself) instead of local copies -- exercises the real
        # transport, e.g. RCCL on a one-GPU machine
        self.p2p_to_self = bool(p2p_to_self)
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        self.nx, self.ny_glob = int(nx), int(ny_glob)
        self.scene, self.scheme = scene, scheme
        self.omega, self.rho, self.u = float(omega), float(rho), float(u)
        self.device = torch.device(device)
        self.n, self.y_below = slab_rows(self.ny_glob, self.world, self.rank)
        if self.n < 1:
            raise ValueError(f"rank {self.rank} owns no rows: NY={ny_glob} < world size {self.world}")
        if backend is None:
            if self.device.type != "cuda":
                raise RuntimeError("SlabSimulation needs a HIP device; there is no CPU path")
            backend = HipBackend()
        self.backend = backend
        self.periodic = scene == SHEAR_WAVE
        # neighbours along y; None = wall (lid scene at the first / last rank)
        self.lower: Optional[int] = (self.rank - 1) % self.world if (self.periodic or self.rank > 0) else None
        self.upper: Optional[int] = (self.rank + 1) % self.world if (self.periodic or self.rank < self.world - 1) else None

        n = self.n
        # Ghost depth G.  halo=None: ask the library how many steps one launch should take for a
        # slab of the smallest rank's height (every rank must arrive at the same answer); a depth
        # above 1 selects the deep-halo schedule and needs slabs of at least 2G rows.
        if halo is None:
            probe = torch.empty((9, 1, self.nx), dtype=dtype, device=self.device)
            n_min = self.ny_glob // self.world
            halo = int(backend.plan_steps(probe, n_min, 1 << 30, scene == LID_DRIVEN)) if hasattr(backend, "plan_steps") else 1
            if n_min < 2 * halo:
                halo = 1
        self.G = G = int(halo)
        if G < 1:
            raise ValueError(f"halo depth must be positive (got {halo})")
        if G > 1 and self.ny_glob // self.world < 2 * G:
            raise ValueError(f"halo depth {G} needs slabs of at least {2 * G} rows")
        self.deep = G > 1
        self.f = torch.empty((9, n + 2 * G, self.nx), dtype=dtype, device=self.device)
        self.buf = torch.empty_like(self.f)
        # ghost rows start defined (a wall-side ghost row is never used by the lid kernels, a
        # periodic one is overwritten by the first exchange)
        self.f[:, :G].zero_(); self.f[:, n + G:].zero_()
        self.buf[:, :G].zero_(); self.buf[:, n + G:].zero_()
        self.blocks = 0     # exchanges so far = launches of the interior sweep
        self.launch_steps = []   # steps taken by each of the most recent blocks (bounded)
        self._stage = None  # deep-halo staging buffers, allocated at the first exchange
        self._max = torch.empty(2, dtype=dtype, device=self.device)
        self.t = 0

        self.cuda = self.device.type == "cuda"
        # RCCL ("nccl") orders its transfers on the stream they are issued from.  Any other transport
        # (gloo, used for rehearsals and tests) is driven by host threads that know nothing about
        # streams, so device work must be drained by the host before every exchange.
        have_group = dist.is_available() and dist.is_initialized()
        if self.p2p_to_self and not have_group:
            raise RuntimeError("p2p_to_self needs an initialised process group")
        self.stream_ordered = (not have_group or not self.cuda or dist.get_backend(self.group) == "nccl")
        self.overlap = bool(overlap) and self.cuda
        if self.cuda:
            self.compute_stream = torch.cuda.current_stream(self.device)   # interior sweeps
            # High priority: the edge sweeps and the exchange are short and everybody waits for them.
            # It also keeps this stream off the hardware queue of the interior stream (HIP gives
            # priority levels their own queues); two streams multiplexed on one hardware queue are
            # processed in submission order and the interior sweep would queue behind the exchange.
            # Ranks that share a device with another rank (rehearsals) take it at default priority.
            self.shares_device = ranks_share_device(self.device, self.group) if have_group else False
            self.comm_stream = (_edge_stream(self.device, high_priority=not self.shares_device)
                                if self.overlap else self.compute_stream)
            # per step parity (two of each, so that "the previous step's" event can still be waited
            # for after this step's work has been submitted):
            self._edges_done = [torch.cuda.Event(), torch.cuda.Event()]      # edge rows are in buf
            self._interior_done = [torch.cuda.Event(), torch.cuda.Event()]   # interior rows are in buf
            self._halo_done = torch.cuda.Event()       # the last exchange has completed
        self._halo_pending = False

        if scene == LID_DRIVEN:
            backend.init_rest(self.f, self.rho)
        elif self.deep:
            backend.init_shearwave(self.f, self.rho, self.u, self.y_below, self.ny_glob, halo=G)
        else:
            backend.init_shearwave(self.f, self.rho, self.u, self.y_below, self.ny_glob)
        if self.overlap:
            # the edge stream's first sweeps and exchanges come after the initialisation, whatever
            # the schedule (the initialisers run on the compute stream)
            self._interior_done[1].record(self.compute_stream)
            self.comm_stream.wait_event(self._interior_done[1])
        if self.deep or self.scheme == "pull":
            # pull reads ghost rows of f: fill them from the neighbours' initial edge rows
            if self.deep:
                self._exchange_deep(self.f)
            else:
                self._exchange(self.f, pull=True)
            self._join_halo()

    # ------------------------------------------------------------------ exchange -------------
    def _exchange(self, a: torch.Tensor, pull: bool) -> None:
        """Halo-row exchange on array `a`, issued on the edge/communication stream; the caller has
        ordered that stream after whatever produced the rows to be sent.

        pull: my edge rows -> the neighbours' ghost rows.   send a[DOWN,1] down, a[UP,n] up;
              receive a[UP,0] from below, a[DOWN,n+1] from above.
        push: my ghost rows -> the neighbours' edge rows.   send a[DOWN,0] down, a[UP,n+1] up;
              receive a[UP,1] from below, a[DOWN,n] from above.
        """
        n = self.n
        if pull:
            s_dn, s_up, r_dn, r_up = 1, n, 0, n + 1
        else:
            s_dn, s_up, r_dn, r_up = 0, n + 1, 1, n

        # x range that travels, per channel.  Periodic scenes and pull: the whole row.  Lid-driven
        # push: a population moving in -x never reaches x=nx-1 of the next row and one moving in +x
        # never reaches x=0 (their would-be sources lie behind the side walls); those two slots of
        # the destination row hold the bounced-back populations of the wall nodes themselves
        # (reference pull-push.cpp:520-528) and must not be overwritten by the exchange.
        def xs(q):
            if pull or self.periodic:
                return slice(None)
            if q in (6, 7):   # cx = -1
                return slice(0, self.nx - 1)
            if q in (5, 8):   # cx = +1
                return slice(1, self.nx)
            return slice(None)
        ctx = torch.cuda.stream(self.comm_stream) if self.cuda else _NullCtx()
        with ctx:
            if self.world == 1 and not self.p2p_to_self:
                if self.lower is not None:  # periodic: the neighbour is this rank itself
                    for q in DOWN_CHANNELS:
                        a[q, r_up, xs(q)].copy_(a[q, s_dn, xs(q)])
                    for q in UP_CHANNELS:
                        a[q, r_dn, xs(q)].copy_(a[q, s_up, xs(q)])
            else:
                # Matching between one pair of ranks is by issue order (RCCL has no tags): what I
                # send down is what the lower rank receives "from above", so sends go [down, up] and
                # receives [from above, from below]; with two ranks both neighbours are the same
                # peer and this order is what keeps the two directions apart.
                p2p = []
                if self.lower is not None:
                    for q in DOWN_CHANNELS:
                        p2p.append(("send", a[q, s_dn, xs(q)], self.lower, q))
                if self.upper is not None:
                    for q in UP_CHANNELS:
                        p2p.append(("send", a[q, s_up, xs(q)], self.upper, q))
                if self.upper is not None:
                    for q in DOWN_CHANNELS:
                        p2p.append(("recv", a[q, r_up, xs(q)], self.upper, q))
                if self.lower is not None:
                    for q in UP_CHANNELS:
                        p2p.append(("recv", a[q, r_dn, xs(q)], self.lower, q))
                self._transfer(p2p)
            if self.cuda and self.overlap:
                self._halo_done.record(self.comm_stream)
        self._halo_pending = True

    def _exchange_deep(self, a: torch.Tensor) -> None:
        """Deep-halo exchange on array `a` (edge/communication stream): my G lowest and G highest
        rows, all nine populations, become the neighbours' ghost rows.  The nine [G, NX] blocks of
        one edge are gathered into one contiguous staging buffer, so an exchange is two sends and
        two receives whatever G is (every message costs the transport a fixed overhead)."""
        n, G = self.n, self.G
        lo_own, hi_own = slice(G, 2 * G), slice(n, n + G)          # rows I send down / up
        lo_ghost, hi_ghost = slice(0, G), slice(n + G, n + 2 * G)   # rows I receive from below / above
        ctx = torch.cuda.stream(self.comm_stream) if self.cuda else _NullCtx()
        with ctx:
            if self.world == 1 and not self.p2p_to_self:
                if self.lower is not None:  # periodic: the neighbour is this rank itself
                    a[:, hi_ghost].copy_(a[:, lo_own])
                    a[:, lo_ghost].copy_(a[:, hi_own])
            else:
                if self._stage is None:
                    self._stage = torch.empty((4, 9, G, self.nx), dtype=a.dtype, device=a.device)
                send_lo, send_hi, recv_lo, recv_hi = self._stage
                # same pairing rule as _exchange: sends [down, up], receives [from above, from below]
                p2p = []
                if self.lower is not None:
                    send_lo.copy_(a[:, lo_own])
                    p2p.append(("send", send_lo, self.lower, 0))
                if self.upper is not None:
                    send_hi.copy_(a[:, hi_own])
                    p2p.append(("send", send_hi, self.upper, 0))
                if self.upper is not None:
                    p2p.append(("recv", recv_hi, self.upper, 0))
                if self.lower is not None:
                    p2p.append(("recv", recv_lo, self.lower, 0))
                self._transfer(p2p)
                if self.upper is not None:
                    a[:, hi_ghost].copy_(recv_hi)
                if self.lower is not None:
                    a[:, lo_ghost].copy_(recv_lo)
            if self.cuda and self.overlap:
                self._halo_done.record(self.comm_stream)
        self._halo_pending = True

    def _transfer(self, items) -> None:
        """One grouped send/recv; items are ("send" | "recv", tensor, group rank of the peer, tag)
        in posting order.  A transport that is not stream-ordered (gloo, the rehearsal transport)
        gets host tensors.  Handed device tensors, gloo writes the received bytes into device
        memory with host stores through the PCIe aperture; nothing orders those stores against a
        kernel launched afterwards (the request completes when the host has issued them, not when
        they have landed), and a later sweep now and then read a few thousand stale values.  A
        staged receive ends in an ordinary host-to-device copy on the stream instead."""
        if not items:
            return
        staged = self.cuda and not self.stream_ordered
        if self.cuda and not self.stream_ordered:
            (self.comm_stream if self.overlap else self.compute_stream).synchronize()
        bufs = []
        for kind, t, _, _ in items:
            if not staged:
                bufs.append(t)
            elif kind == "send":
                bufs.append(t.cpu())
            else:
                bufs.append(torch.empty(t.shape, dtype=t.dtype, device="cpu"))
        ops = [dist.P2POp(dist.isend if kind == "send" else dist.irecv, b, self._peer(peer), self.group, tag=tag)
               for (kind, _, peer, tag), b in zip(items, bufs)]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if staged:
            for (kind, t, _, _), b in zip(items, bufs):
                if kind == "recv":
                    t.copy_(b)

    def _sweep_fused(self, y_lo: int, y_hi: int, k: int) -> None:
        if y_lo >= y_hi:
            return
        b = self.backend
        if self.scene == LID_DRIVEN:
            b.lid_fused(self.scheme, self.f, self.buf, self.omega, self.rho, self.u, y_lo, y_hi,
                        self.y_below - self.G, self.ny_glob, k)
        else:
            b.periodic_fused(self.scheme, self.f, self.buf, self.omega, y_lo, y_hi, k)

    def _block(self, k: int) -> None:
        """Advance k <= G time steps: edge regions (G rows at either end) and the exchange on the
        edge stream, the interior on the compute stream.  The interior of a block reads owned rows
        only, so it never waits for an exchange; it needs the previous block's edge rows, and the
        edge regions need the previous block's interior rows and exchange."""
        n, G = self.n, self.G
        if self.overlap:
            c = self.blocks & 1
            p = c ^ 1
            self.comm_stream.wait_event(self._interior_done[p])
            self.compute_stream.wait_event(self._edges_done[p])
            # interior first: see step()
            self._sweep_fused(2 * G, n, k)
            self._interior_done[c].record(self.compute_stream)
            with torch.cuda.stream(self.comm_stream):
                self._sweep_fused(G, 2 * G, k)
                self._sweep_fused(n, n + G, k)
                self._edges_done[c].record(self.comm_stream)
            self._exchange_deep(self.buf)
        else:
            self._sweep_fused(G, 2 * G, k)
            self._sweep_fused(n, n + G, k)
            self._exchange_deep(self.buf)
            self._sweep_fused(2 * G, n, k)
        self.f, self.buf = self.buf, self.f
        self.blocks += 1
        self.launch_steps.append(k)
        del self.launch_steps[:-4096]
        self.t += k

    def _peer(self, group_rank: int) -> int:
        return dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank

    def _join_halo(self) -> None:
        """Make the compute stream wait for the last exchange."""
        if self._halo_pending and self.cuda and self.overlap:
            self.compute_stream.wait_event(self._halo_done)
        self._halo_pending = False

    # ------------------------------------------------------------------ sweeps ---------------
    def _sweep(self, y_lo: int, y_hi: int) -> None:
        if y_lo >= y_hi:
            return
        b, f, buf = self.backend, self.f, self.buf
        if self.scene == LID_DRIVEN:
            args = (f, buf, self.omega, self.rho, self.u, y_lo, y_hi, self.y_below - 1, self.ny_glob)
            (b.pull_lid_rows if self.scheme == "pull" else b.push_lid_rows)(*args)
        elif self.scheme == "pull":
            b.pull_periodic_rows(f, buf, self.omega, y_lo, y_hi)
        else:
            b.push_periodic_rows(f, buf, self.omega, y_lo, y_hi)

    def step(self, nsteps: int = 1) -> None:
        n = self.n
        pull = self.scheme == "pull"
        if self.deep:
            # the fewest blocks of at most G steps, as equal as possible (a short tail block pays a
            # whole pass over memory and a whole exchange for its few steps)
            nsteps = int(nsteps)
            while nsteps > 0:
                blocks = -(-nsteps // self.G)
                k = -(-nsteps // blocks)
                self._block(k)
                nsteps -= k
            return
        for _ in range(nsteps):
            if self.overlap:
                k = self.t & 1
                p = k ^ 1
                # edge stream: needs the previous step's interior rows (and, by stream order, its
                # edge rows and exchange)
                self.comm_stream.wait_event(self._interior_done[p])
                # interior stream: needs the previous step's edge rows (and, by stream order, its
                # interior rows); it never waits for an exchange
                self.compute_stream.wait_event(self._edges_done[p])
                # The interior sweep is submitted first: streams share a few hardware queues that
                # are processed in submission order, and nothing of this step's edge/exchange work
                # may sit in front of the big kernel.
                self._sweep(2, n)
                self._interior_done[k].record(self.compute_stream)
                with torch.cuda.stream(self.comm_stream):
                    self._sweep(1, 2)
                    if n >= 2:
                        self._sweep(n, n + 1)
                    self._edges_done[k].record(self.comm_stream)
                self._exchange(self.buf, pull=pull)
            else:
                self._sweep(1, 2)
                if n >= 2:
                    self._sweep(n, n + 1)
                self._exchange(self.buf, pull=pull)
                self._sweep(2, n)
            self.f, self.buf = self.buf, self.f
            self.t += 1
            self.blocks += 1
            self.launch_steps.append(1)
            del self.launch_steps[:-4096]

    # ------------------------------------------------------------------ observables ----------
    def max_velocity(self) -> float:
        """Global max |u| (reference io.cpp:184-216: local reduce, then MPI_Reduce(MAX))."""
        self._join_halo()
        self.backend.max_velocity_dev(self._max, self.f, self.G, self.G + self.n)
        if self.world > 1:
            if self.cuda and not self.stream_ordered:
                m = self._max[:1].cpu()             # see _transfer
                dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
                return float(m.item())
            dist.all_reduce(self._max[:1], op=dist.ReduceOp.MAX, group=self.group)
        return float(self._max[0].item())

    def interior(self) -> torch.Tensor:
        """This rank's rows without the ghost rows: [9, n, NX] (a view)."""
        self._join_halo()
        return self.f[:, self.G:self.G + self.n]

    def gather_populations(self) -> Optional[torch.Tensor]:
        """Rank 0 gets the global [9, NY, NX] populations (for tests / small outputs)."""
        mine = self.interior().contiguous()
        if self.world == 1:
            return mine
        staged = self.cuda and not self.stream_ordered      # see _transfer
        if self.rank == 0:
            parts = [mine]
            for r in range(1, self.world):
                nr, _ = slab_rows(self.ny_glob, self.world, r)
                part = torch.empty((9, nr, self.nx), dtype=mine.dtype, device="cpu" if staged else self.device)
                dist.recv(part, self._peer(r), group=self.group)
                parts.append(part.to(self.device))
            return torch.cat(parts, dim=1)
        dist.send(mine.cpu() if staged else mine, self._peer(0), group=self.group)
        return None

    def synchronize(self) -> None:
        if self.cuda:
            self._join_halo()
            torch.cuda.synchronize(self.device)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
