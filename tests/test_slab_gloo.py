"""The multi-rank path on CPU ranks (gloo): slab partition, halo-row exchange schedule, global max.

The slab kernels are replaced by the oracle (tests/slab_backend.py), so what is under test is the
host logic of lbm_amd.slab: which rows are exchanged with whom and when.  The gathered multi-rank
populations must equal the single-domain oracle run bit for bit (the reference's own MPI
initialisation offset, main.cpp:141-144, is wrong for ranks > 0 and is deliberately not
reproduced)."""
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


def _worker(rank, world, store_path, cases, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(1)
    os.environ["OMP_NUM_THREADS"] = "2"
    from lbm_amd.slab import SlabSimulation
    from slab_backend import OracleBackend
    dist.init_process_group("gloo", init_method=f"file://{store_path}", rank=rank, world_size=world)
    results = {}
    try:
        for (scene, scheme, tdt, nx, ny, steps, omega, depth) in cases:
            backend = OracleBackend()
            backend.fused_depth = depth
            sim = SlabSimulation(nx, ny, scene=scene, scheme=scheme, dtype=getattr(torch, tdt), omega=omega,
                                 device="cpu", backend=backend)
            assert sim.G == depth and sim.deep == (depth > 1)
            vmax0 = sim.max_velocity()
            sim.step(steps)
            vmax = sim.max_velocity()
            full = sim.gather_populations()
            if rank == 0:
                results[(scene, scheme, tdt, nx, ny, steps, omega, depth)] = (full.numpy().copy(), vmax0, vmax)
        if rank == 0:
            torch.save(results, out_path)
    finally:
        dist.barrier()
        dist.destroy_process_group()


# (scene, scheme, dtype, nx, ny, steps, omega, ghost depth): depth 1 = one halo row exchanged per
# step, depth > 1 = the deep-halo schedule (blocks of `depth` steps, the last block shorter)
CASES = [
    ("shear", "pull", "float32", 16, 13, 25, 0.8, 1),
    ("shear", "push", "float32", 16, 13, 25, 0.8, 1),
    ("shear", "pull", "float64", 12, 10, 40, 1.3, 1),
    ("shear", "push", "float64", 12, 10, 40, 1.3, 1),
    ("lid", "push", "float64", 12, 11, 60, 1.1, 1),
    ("lid", "pull", "float64", 12, 11, 60, 1.1, 1),
    ("lid", "push", "float32", 8, 7, 30, 1.7, 1),
    ("shear", "pull", "float64", 8, 4, 10, 0.5, 1),     # some ranks own a single row
    ("shear", "pull", "float32", 16, 25, 26, 0.8, 4),
    ("shear", "push", "float32", 16, 25, 26, 0.8, 4),
    ("shear", "pull", "float64", 12, 24, 9, 1.3, 4),    # slabs of exactly 2 x depth rows (world 3)
    ("shear", "push", "float64", 12, 19, 17, 1.3, 3),
    ("lid", "push", "float64", 12, 26, 31, 1.1, 4),
    ("lid", "pull", "float64", 12, 26, 31, 1.1, 4),
    ("lid", "push", "float32", 8, 13, 30, 1.7, 2),
    ("shear", "push", "float32", 16, 37, 27, 0.8, 6),
    ("lid", "pull", "float64", 12, 36, 20, 1.1, 6),   # slabs of exactly 2 x depth rows (world 3)
]


@pytest.mark.parametrize("world", [2, 3])
def test_slabs_reproduce_the_single_domain_run(world):
    from oracle import oracle as orc
    with tempfile.TemporaryDirectory() as tmp:
        store, out = os.path.join(tmp, "store"), os.path.join(tmp, "out.pt")
        mp.spawn(_worker, args=(world, store, CASES, out), nprocs=world, join=True)
        results = torch.load(out, weights_only=False)
    for case in CASES:
        scene, scheme, tdt, nx, ny, steps, omega, _depth = case
        got, vmax0, vmax = results[case]
        ndt = np.float32 if tdt == "float32" else np.float64
        want, _ = orc.run(scene, scheme, nx, ny, steps, omega, ndt)
        assert got.shape == want.shape, case
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (case, np.abs(got - want).max())
        f0, _ = orc.run(scene, scheme, nx, ny, 0, omega, ndt)
        assert vmax0 == float(orc.max_velocity(f0)), case
        assert vmax == float(orc.max_velocity(want)), case


def test_slab_rows_partition():
    from lbm_amd.slab import slab_rows
    for ny, world in ((10, 3), (16384, 8), (7, 7), (13, 2), (32768, 8)):
        parts = [slab_rows(ny, world, r) for r in range(world)]
        assert sum(n for n, _ in parts) == ny
        below = 0
        for n, y_below in parts:
            assert y_below == below
            below += n
        assert max(n for n, _ in parts) - min(n for n, _ in parts) <= 1


def test_single_process_world_of_one_uses_self_exchange():
    """World size 1 without a process group: the periodic neighbours are the rank itself."""
    from lbm_amd.slab import SlabSimulation
    from slab_backend import OracleBackend
    from oracle import oracle as orc
    for depth in (1, 4):
        for scene, scheme in (("shear", "pull"), ("shear", "push"), ("lid", "push"), ("lid", "pull")):
            backend = OracleBackend()
            backend.fused_depth = depth
            sim = SlabSimulation(10, 9, scene=scene, scheme=scheme, dtype=torch.float64, omega=0.9,
                                 device="cpu", backend=backend)
            assert sim.G == depth
            sim.step(15)
            want, _ = orc.run(scene, scheme, 10, 9, 15, 0.9, np.float64)
            assert np.array_equal(sim.interior().numpy(), want), (scene, scheme, depth)
            assert sim.max_velocity() == float(orc.max_velocity(want))


def test_cpu_device_without_backend_is_refused():
    from lbm_amd.slab import SlabSimulation
    with pytest.raises(RuntimeError):
        SlabSimulation(8, 8, device="cpu")
