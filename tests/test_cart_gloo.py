"""The 2-D Cartesian decomposition (lbm_amd.cart) on CPU ranks (gloo).

The block kernels are replaced by the oracle (tests/slab_backend.py), so what is under test is the
host logic: the rank grid, the block extents, which populations go to which neighbour in which
phase, the corner values riding with the rows.  The gathered populations must equal the
single-domain oracle run bit for bit."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# (dtype, nx, ny, steps, omega); extents that do not divide by the grid leave remainders
CASES = [
    ("float32", 16, 12, 25, 0.8),
    ("float64", 13, 11, 30, 1.3),
    ("float64", 9, 9, 12, 1.9),      # blocks of two nodes on a grid of four
]


def _worker(rank, world, dims, store_path, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    os.environ["OMP_NUM_THREADS"] = "1"
    from lbm_amd.cart import CartSimulation
    from slab_backend import OracleCartBackend
    dist.init_process_group("gloo", init_method=f"file://{store_path}", rank=rank, world_size=world)
    results = {}
    try:
        for (tdt, nx, ny, steps, omega) in CASES:
            for scheme in ("pull", "push"):
                sim = CartSimulation(nx, ny, dtype=getattr(torch, tdt), omega=omega, device="cpu",
                                     backend=OracleCartBackend(), dims=dims, scheme=scheme)
                vmax0 = sim.max_velocity()
                sim.step(steps)
                vmax = sim.max_velocity()
                full = sim.gather_populations()
                if rank == 0:
                    results[(tdt, nx, ny, steps, omega, scheme)] = (full.numpy().copy(), vmax0, vmax)
        if rank == 0:
            torch.save(results, out_path)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_dims_and_extents():
    from lbm_amd.cart import block_extent, dims_create
    assert [dims_create(w) for w in (1, 2, 3, 4, 6, 8, 12, 16)] == [(1, 1), (2, 1), (3, 1), (2, 2), (3, 2), (4, 2),
                                                                     (4, 3), (4, 4)]
    parts = [block_extent(11, 3, c) for c in range(3)]
    assert parts == [(4, 0), (4, 4), (3, 8)]
    with pytest.raises(ValueError):
        dims_create(0)


@pytest.mark.parametrize("world,dims", [(2, None), (4, None), (3, (1, 3)), (4, (4, 1)), (6, None)])
def test_blocks_reproduce_the_single_domain_run(world, dims):
    from oracle import oracle as orc
    with tempfile.TemporaryDirectory() as tmp:
        store, out = os.path.join(tmp, "store"), os.path.join(tmp, "out.pt")
        mp.spawn(_worker, args=(world, dims, store, out), nprocs=world, join=True)
        results = torch.load(out, weights_only=False)
    for (tdt, nx, ny, steps, omega, scheme), (full, vmax0, vmax) in results.items():
        dtype = getattr(np, tdt)
        want, _ = orc.run("shear", scheme, nx, ny, steps, omega, dtype)
        assert full.dtype == want.dtype and full.shape == want.shape
        bits = np.uint32 if dtype == np.float32 else np.uint64
        assert (full.view(bits) == want.view(bits)).all(), (world, dims, tdt, nx, ny, scheme)
        start = orc.alloc(nx, ny, dtype)
        orc.init_shearwave(start, dtype(1.0), dtype(0.1))
        assert vmax0 == float(orc.max_velocity(start))
        assert vmax == float(orc.max_velocity(want))


# blocks of several steps per exchange: (dtype, nx, ny, steps, omega, frame depth); the smallest block
# of every grid below has at least `depth` owned rows and columns
DEEP_CASES = [
    ("float32", 24, 18, 25, 0.8, 3),     # odd local row lengths get the extra column
    ("float64", 17, 19, 30, 1.3, 4),
    ("float32", 16, 12, 13, 1.9, 2),
    ("float64", 20, 21, 16, 0.6, None),  # depth chosen by the backend's plan (5)
]


def _worker_deep(rank, world, dims, store_path, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    os.environ["OMP_NUM_THREADS"] = "1"
    from lbm_amd.cart import CartSimulation
    from slab_backend import OracleCartBackend
    dist.init_process_group("gloo", init_method=f"file://{store_path}", rank=rank, world_size=world)
    results = {}
    try:
        for (tdt, nx, ny, steps, omega, depth) in DEEP_CASES:
            for scheme in ("pull", "push"):
                backend = OracleCartBackend()
                backend.fused_depth = 5
                sim = CartSimulation(nx, ny, dtype=getattr(torch, tdt), omega=omega, device="cpu",
                                     backend=backend, dims=dims, scheme=scheme, halo=depth)
                assert sim.deep and sim.G == (depth or 5) and sim.f.shape[2] % 2 == 0
                vmax0 = sim.max_velocity()
                sim.step(steps)
                assert sim.t == steps
                vmax = sim.max_velocity()
                full = sim.gather_populations()
                if rank == 0:
                    results[(tdt, nx, ny, steps, omega, depth, scheme)] = (full.numpy().copy(), vmax0, vmax)
        if rank == 0:
            torch.save(results, out_path)
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dims", [(2, None), (4, None), (3, (1, 3)), (6, None)])
def test_deep_frames_reproduce_the_single_domain_run(world, dims):
    """Blocks of up to `depth` steps per exchange on frames `depth` nodes deep, all nine populations,
    columns then rows: bit for bit the single-domain run, pull and push."""
    from oracle import oracle as orc
    with tempfile.TemporaryDirectory() as tmp:
        store, out = os.path.join(tmp, "store"), os.path.join(tmp, "out.pt")
        mp.spawn(_worker_deep, args=(world, dims, store, out), nprocs=world, join=True)
        results = torch.load(out, weights_only=False)
    assert len(results) == 2 * len(DEEP_CASES)
    for (tdt, nx, ny, steps, omega, depth, scheme), (full, vmax0, vmax) in results.items():
        dtype = getattr(np, tdt)
        want, _ = orc.run("shear", scheme, nx, ny, steps, omega, dtype)
        bits = np.uint32 if dtype == np.float32 else np.uint64
        assert (full.view(bits) == want.view(bits)).all(), (world, dims, tdt, nx, ny, depth, scheme)
        start = orc.alloc(nx, ny, dtype)
        orc.init_shearwave(start, dtype(1.0), dtype(0.1))
        assert vmax0 == float(orc.max_velocity(start))
        assert vmax == float(orc.max_velocity(want))


# the lid-driven cavity cut in both directions: (dtype, nx, ny, steps, omega, frame depth)
CAVITY_CASES = [
    ("float32", 24, 18, 25, 1.1, 3),
    ("float64", 17, 19, 30, 1.7, 4),     # odd rows: the extra column goes beyond a frame, never to a wall
    ("float64", 13, 11, 20, 0.9, 1),     # frames of one node, one step per exchange
    ("float32", 20, 21, 16, 0.6, None),  # depth chosen by the backend's plan (5)
]


def _worker_cavity(rank, world, dims, store_path, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    os.environ["OMP_NUM_THREADS"] = "1"
    from lbm_amd.cart import CartSimulation
    from slab_backend import OracleCartBackend
    dist.init_process_group("gloo", init_method=f"file://{store_path}", rank=rank, world_size=world)
    results = {}
    try:
        for (tdt, nx, ny, steps, omega, depth) in CAVITY_CASES:
            for scheme in ("pull", "push"):
                backend = OracleCartBackend()
                backend.fused_depth = 5
                sim = CartSimulation(nx, ny, dtype=getattr(torch, tdt), omega=omega, device="cpu", backend=backend,
                                     dims=dims, scheme=scheme, halo=depth, scene="lid")
                assert sim.G == (depth or 5)
                if tdt == "float32" and sim.dims[0] > 1:
                    assert sim.f.shape[2] % 2 == 0
                sim.step(steps)
                vmax = sim.max_velocity()
                full = sim.gather_populations()
                if rank == 0:
                    results[(tdt, nx, ny, steps, omega, depth, scheme)] = (full.numpy().copy(), vmax)
        if rank == 0:
            torch.save(results, out_path)
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dims", [(2, None), (4, None), (3, (1, 3)), (6, None), (3, (3, 1))])
def test_cavity_blocks_reproduce_the_single_domain_run(world, dims):
    """The lid-driven cavity on a grid of ranks (the reference refuses it under MPI, main.cpp:317-319):
    frames only towards neighbours, walls at the array ends of the blocks that touch them; bit for
    bit the single-domain run, pull and push."""
    from oracle import oracle as orc
    with tempfile.TemporaryDirectory() as tmp:
        store, out = os.path.join(tmp, "store"), os.path.join(tmp, "out.pt")
        mp.spawn(_worker_cavity, args=(world, dims, store, out), nprocs=world, join=True)
        results = torch.load(out, weights_only=False)
    assert len(results) == 2 * len(CAVITY_CASES)
    for (tdt, nx, ny, steps, omega, depth, scheme), (full, vmax) in results.items():
        dtype = getattr(np, tdt)
        want, _ = orc.run("lid", scheme, nx, ny, steps, omega, dtype)
        bits = np.uint32 if dtype == np.float32 else np.uint64
        assert (full.view(bits) == want.view(bits)).all(), (world, dims, tdt, nx, ny, depth, scheme)
        assert vmax == float(orc.max_velocity(want))


def test_cavity_on_one_rank_has_no_frames():
    from oracle import oracle as orc
    from lbm_amd.cart import CartSimulation
    from slab_backend import OracleCartBackend
    for scheme in ("pull", "push"):
        sim = CartSimulation(9, 7, dtype=torch.float64, omega=1.2, device="cpu", backend=OracleCartBackend(),
                             scheme=scheme, scene="lid", halo=3)
        assert tuple(sim.f.shape) == (9, 7, 9) and (sim.x0, sim.y0) == (0, 0)
        sim.step(11)
        want, _ = orc.run("lid", scheme, 9, 7, 11, 1.2, np.float64)
        assert (sim.gather_populations().numpy().view(np.uint64) == want.view(np.uint64)).all(), scheme


def test_deep_frames_limits():
    from lbm_amd.cart import CartSimulation
    from slab_backend import OracleCartBackend
    with pytest.raises(ValueError, match="halo depth"):
        CartSimulation(9, 6, device="cpu", backend=OracleCartBackend(), halo=7)     # more than the block has rows
    with pytest.raises(ValueError, match="halo depth"):
        CartSimulation(9, 6, device="cpu", backend=OracleCartBackend(), halo=0)
    b = OracleCartBackend()
    b.fused_depth = 8
    sim = CartSimulation(9, 6, device="cpu", backend=b)      # planned depth is capped by the block: 6 rows
    assert sim.G == 6 and sim.deep


def test_single_rank_wraps_onto_itself():
    """No process group at all: every neighbour is the rank itself."""
    from oracle import oracle as orc
    from lbm_amd.cart import CartSimulation
    from slab_backend import OracleCartBackend
    for scheme in ("pull", "push"):
        sim = CartSimulation(9, 6, dtype=torch.float64, omega=1.1, device="cpu", backend=OracleCartBackend(), scheme=scheme)
        sim.step(17)
        want, _ = orc.run("shear", scheme, 9, 6, 17, 1.1, np.float64)
        assert (sim.gather_populations().numpy().view(np.uint64) == want.view(np.uint64)).all(), scheme
    with pytest.raises(RuntimeError):
        CartSimulation(9, 6, device="cpu")       # the default kernels are HIP-only
    with pytest.raises(ValueError):
        CartSimulation(9, 1, device="cpu", backend=OracleCartBackend())
