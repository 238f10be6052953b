"""The 2-D Cartesian decomposition (lbm_amd.cart) on the GPU: the HIP inner / outer sweeps on a
block with a halo ring, the packing and the second stream, and the multi-step sweeps on blocks
with deep frames -- first with the rank as its own neighbour in both directions, then four and
six ranks sharing cuda:0 over gloo (RCCL refuses several ranks on one GPU).

The file name sorts behind the single-process GPU tests on purpose: the tests that start several
ranks on one device run last (before the allocation-free entry point's), so that a rank that stalls
there fails its own test under `pytest -x` without hiding the rest of the suite."""
import faulthandler
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as orc
from test_parity_gpu import assert_bits_equal

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("scheme", ["pull", "push"])
@pytest.mark.parametrize("halo", [1, None, 3])
def test_one_rank_equals_the_oracle(dtype, overlap, scheme, halo):
    """halo 1: the reference's scheme (inner sweep || exchange, outer ring); None / 3: blocks of
    several steps per exchange on deep frames (the depth the library plans, or three)."""
    from lbm_amd.cart import CartSimulation
    npdt = np.float32 if dtype == torch.float32 else np.float64
    for nx, ny, steps, omega in ((64, 48, 40, 0.8), (131, 37, 25, 1.4), (2, 2, 9, 1.0), (1024, 300, 12, 0.5)):
        if halo == 3 and min(nx, ny) < 3:
            with pytest.raises(ValueError):
                CartSimulation(nx, ny, dtype=dtype, omega=omega, overlap=overlap, scheme=scheme, halo=halo)
            continue
        sim = CartSimulation(nx, ny, dtype=dtype, omega=omega, overlap=overlap, scheme=scheme, halo=halo)
        assert sim.deep == (halo != 1 and sim.G > 1) and (halo is None or sim.G == halo)
        if halo is None and min(nx, ny) >= 8:
            assert sim.deep, "the planned depth should select deep frames on blocks this large"
        sim.step(steps)
        want, _ = orc.run("shear", scheme, nx, ny, steps, omega, npdt)
        assert_bits_equal(sim.gather_populations().cpu().numpy(), want, f"cart 1x1 {scheme} {nx}x{ny}")
        assert sim.max_velocity() == float(orc.max_velocity(want))
        assert sim.t == steps


CASES = [("float32", 256, 96, 30, 0.8), ("float64", 67, 45, 40, 1.3), ("float32", 1030, 517, 12, 0.5)]


def _worker(rank, world, store_path, out_path):
    sys.path.insert(0, ROOT)
    # a rank that stops making progress prints its stacks and exits: the test fails, the suite goes on
    faulthandler.dump_traceback_later(100, exit=True)
    from lbm_amd.cart import CartSimulation
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{store_path}", rank=rank, world_size=world)
    results = {}
    try:
        for case in CASES:
            tdt, nx, ny, steps, omega = case
            for scheme, halo in (("pull", 1), ("push", 1), ("pull", None), ("push", None), ("pull", 4)):
                sim = CartSimulation(nx, ny, dtype=getattr(torch, tdt), omega=omega, device="cuda:0", scheme=scheme,
                                     halo=halo)
                sim.step(steps)
                vmax = sim.max_velocity()
                full = sim.gather_populations()
                if rank == 0:
                    results[case + (scheme, halo)] = (full.cpu().numpy(), vmax, sim.dims, sim.G)
        for case in CAVITY_CASES:
            tdt, nx, ny, steps, omega = case
            for scheme, halo in (("pull", None), ("push", None), ("push", 3), ("pull", 1)):
                sim = CartSimulation(nx, ny, dtype=getattr(torch, tdt), omega=omega, device="cuda:0", scheme=scheme,
                                     halo=halo, scene="lid")
                sim.step(steps)
                vmax = sim.max_velocity()
                full = sim.gather_populations()
                if rank == 0:
                    results[("lid",) + case + (scheme, halo)] = (full.cpu().numpy(), vmax, sim.dims, sim.G)
        if rank == 0:
            torch.save(results, out_path)
    finally:
        dist.barrier()
        dist.destroy_process_group()


# the lid-driven cavity cut in both directions: (dtype, nx, ny, steps, omega)
CAVITY_CASES = [("float32", 256, 96, 30, 1.1), ("float64", 67, 45, 40, 1.7), ("float32", 1030, 517, 12, 0.5)]


@pytest.mark.parametrize("world", [4, 6])
def test_ranks_sharing_one_gpu_reproduce_the_single_domain_run(world, ranks_share_one_gpu):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "out.pt")
        mp.spawn(_worker, args=(world, os.path.join(tmp, "store"), out), nprocs=world, join=True)
        results = torch.load(out, weights_only=False)
    assert len(results) == 5 * len(CASES) + 4 * len(CAVITY_CASES)
    for key, (full, vmax, dims, depth) in results.items():
        scene = "shear"
        if key[0] == "lid":
            scene, key = "lid", key[1:]
        tdt, nx, ny, steps, omega, scheme, halo = key
        assert dims == ((2, 2) if world == 4 else (3, 2))
        assert depth == halo if halo is not None else depth > 1
        want, _ = orc.run(scene, scheme, nx, ny, steps, omega, getattr(np, tdt))
        assert_bits_equal(full, want, f"cart {dims} {scene} {scheme} {tdt} {nx}x{ny} frame depth {depth}")
        assert vmax == float(orc.max_velocity(want))


def _worker_rccl_self(rank, world, port, out_dir):
    """World of one over RCCL: along both axes the neighbour is this rank itself, and every halo goes
    through the grouped send/recv that a multi-GPU run uses, on the stream the sweeps run on."""
    sys.path.insert(0, ROOT)
    faulthandler.dump_traceback_later(100, exit=True)
    from lbm_amd.cart import CartSimulation
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    ok = True
    try:
        for scheme, tdt, ndt, nx, ny, halo in (("pull", torch.float32, np.float32, 1024, 300, 1),
                                               ("push", torch.float32, np.float32, 1024, 300, 1),
                                               ("pull", torch.float32, np.float32, 1024, 300, None),
                                               ("push", torch.float64, np.float64, 515, 130, None),
                                               ("pull", torch.float64, np.float64, 515, 130, 3),
                                               ("push", torch.float32, np.float32, 1030, 64, 6)):
            steps, omega = 19, 0.7
            sim = CartSimulation(nx, ny, dtype=tdt, omega=omega, device="cuda:0", scheme=scheme, halo=halo,
                                 p2p_to_self=True)
            assert sim.stream_ordered and sim.deep == (halo != 1)
            sim.step(steps)
            vmax = sim.max_velocity()
            want, _ = orc.run("shear", scheme, nx, ny, steps, omega, ndt)
            same = np.array_equal(sim.interior().cpu().numpy().view(np.uint8), want.view(np.uint8))
            ok = ok and same and vmax == float(orc.max_velocity(want))
        with open(os.path.join(out_dir, "result.txt"), "w") as fh:
            fh.write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


def test_rccl_transport_with_self_as_every_neighbour():
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker_rccl_self, args=(1, 29547, tmp), nprocs=1, join=True)
        assert open(os.path.join(tmp, "result.txt")).read() == "ok"


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("scheme", ["pull", "push"])
def test_cavity_on_one_rank_equals_the_oracle(dtype, scheme):
    """One rank: no neighbours, no frames -- the block is the cavity."""
    from lbm_amd.cart import CartSimulation
    npdt = np.float32 if dtype == torch.float32 else np.float64
    # rows of an odd length: fp32 plans one step per launch there unless the arithmetic is the strict one
    for nx, ny, steps, omega, halo in ((64, 48, 40, 0.8, None), (131, 37, 25, 1.4, 1), (131, 37, 25, 1.4, None),
                                       (1024, 300, 12, 0.5, None)):
        sim = CartSimulation(nx, ny, dtype=dtype, omega=omega, scheme=scheme, scene="lid", halo=halo)
        assert tuple(sim.f.shape) == (9, ny, nx)
        sim.step(steps)
        want, _ = orc.run("lid", scheme, nx, ny, steps, omega, npdt)
        assert_bits_equal(sim.gather_populations().cpu().numpy(), want, f"cart 1x1 cavity {scheme} {nx}x{ny}")
        assert sim.max_velocity() == float(orc.max_velocity(want))


@pytest.mark.parametrize("scene", ["shear", "lid"])
def test_blocks_follow_the_process_wide_arithmetic(scene):
    """The decomposition calls the same entry points as the single-device loop: with the contracted
    arithmetic switched on, a block on deep frames holds the bits of the contracted oracle."""
    from lbm_amd import ops
    from lbm_amd.cart import CartSimulation
    with orc.arithmetic("contracted"), ops.arithmetic("contracted"):
        for scheme in ("pull", "push"):
            sim = CartSimulation(1024, 300, dtype=torch.float32, omega=0.9, scheme=scheme, scene=scene)
            assert sim.deep and sim.G >= 5      # what the library plans for a lattice of this size
            sim.step(21)
            want, _ = orc.run(scene, scheme, 1024, 300, 21, 0.9, np.float32)
            assert_bits_equal(sim.gather_populations().cpu().numpy(), want, f"cart contracted {scene} {scheme}")
            # fp32 rows of an odd length: the contracted sweeps take them one step per launch, and a
            # cavity without a neighbour along x keeps its own row length -- the plan must say 1
            sim = CartSimulation(131, 37, dtype=torch.float32, omega=1.4, scheme=scheme, scene=scene)
            if scene == "lid":
                assert sim.G == 1
            sim.step(25)
            want, _ = orc.run(scene, scheme, 131, 37, 25, 1.4, np.float32)
            assert_bits_equal(sim.gather_populations().cpu().numpy(), want, f"cart contracted {scene} {scheme} odd rows")
    assert ops.get_arithmetic() == "unrolled" and orc.get_arithmetic() == "unrolled"
