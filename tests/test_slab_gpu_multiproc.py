"""Two ranks sharing cuda:0 over gloo: the GPU slab path (HIP kernels, second stream, events) under
a real multi-process exchange.  RCCL refuses two ranks on one GPU, so the transport here is gloo;
what is exercised is everything else of lbm_amd.slab on the device."""
import faulthandler
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# A rank that stops making progress prints every thread's stack and exits after this many seconds,
# so a stall fails its test with the place it stalled in instead of hanging the suite.
STALL_SECONDS = 100

CASES = [
    ("shear", "pull", "float32", 256, 37, 30, 0.8),
    ("shear", "push", "float32", 256, 37, 30, 0.8),
    ("shear", "pull", "float64", 64, 24, 40, 1.3),
    ("lid", "push", "float64", 64, 25, 50, 1.1),
    ("lid", "pull", "float32", 128, 27, 50, 1.7),
]


def _worker(rank, world, store_path, out_path):
    sys.path.insert(0, ROOT)
    faulthandler.dump_traceback_later(STALL_SECONDS, exit=True)
    from lbm_amd.slab import SlabSimulation
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{store_path}", rank=rank, world_size=world)
    results = {}
    try:
        for case in CASES:
            scene, scheme, tdt, nx, ny, steps, omega = case
            for halo in (1, 4, 6):   # one halo row per step / deep halos, blocks of four or six fused steps
                sim = SlabSimulation(nx, ny, scene=scene, scheme=scheme, dtype=getattr(torch, tdt), omega=omega,
                                     device="cuda:0", halo=halo)
                sim.step(steps)
                vmax = sim.max_velocity()
                full = sim.gather_populations()
                if rank == 0:
                    results[case + (halo,)] = (full.cpu().numpy(), vmax)
        if rank == 0:
            torch.save(results, out_path)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _worker_big(rank, world, store_path, out_dir):
    """Slabs large enough that the interior sweep really overlaps the exchange.  The shear wave is
    x-invariant, so each rank checks its rows against an 8-column oracle run of the global lattice."""
    sys.path.insert(0, ROOT)
    faulthandler.dump_traceback_later(STALL_SECONDS, exit=True)
    from lbm_amd.slab import SlabSimulation
    from oracle import oracle as orc
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{store_path}", rank=rank, world_size=world)
    ok = True
    try:
        for scheme, overlap, halo in (("pull", True, 1), ("push", True, 1), ("pull", False, 1),
                                      ("pull", True, 4), ("push", True, 4), ("push", False, 4),
                                      ("pull", True, None), ("push", True, 6)):
            nx, ny, steps, omega = 8192, 1024, 14, 0.5
            sim = SlabSimulation(nx, ny, scene="shear", scheme=scheme, dtype=torch.float32, omega=omega,
                                 device="cuda:0", overlap=overlap, halo=halo)
            sim.step(steps)
            vmax = sim.max_velocity()
            want, _ = orc.run("shear", scheme, 8, ny, steps, omega, np.float32)
            col = torch.from_numpy(np.ascontiguousarray(want[:, sim.y_below:sim.y_below + sim.n, :1])).cuda()
            same = bool((sim.interior().view(torch.int32) == col.view(torch.int32)).all())
            ok = ok and same and vmax == float(orc.max_velocity(want))
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as fh:
            fh.write("ok" if ok else "mismatch")
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_four_ranks_with_real_overlap_work(ranks_share_one_gpu):
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker_big, args=(4, os.path.join(tmp, "store"), tmp), nprocs=4, join=True)
        for r in range(4):
            assert open(os.path.join(tmp, f"rank{r}.txt")).read() == "ok", f"rank {r}"


def _worker_rccl_self(rank, world, port, out_dir):
    """World of one over RCCL: the periodic neighbours are this rank itself, and the halo rows go
    through exactly the grouped send/recv on the second stream that a multi-GPU run uses."""
    sys.path.insert(0, ROOT)
    faulthandler.dump_traceback_later(STALL_SECONDS, exit=True)
    from lbm_amd.slab import SlabSimulation
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    ok = True
    try:
        for scheme, tdt, ndt, nx, ny, halo in (("pull", torch.float32, np.float32, 8192, 512, 1),
                                               ("push", torch.float32, np.float32, 8192, 512, 1),
                                               ("pull", torch.float64, np.float64, 4096, 256, 1),
                                               ("push", torch.float64, np.float64, 1030, 64, 1),
                                               ("pull", torch.float32, np.float32, 8192, 512, 4),
                                               ("push", torch.float32, np.float32, 8192, 512, 4),
                                               ("pull", torch.float64, np.float64, 4096, 256, 4),
                                               ("push", torch.float64, np.float64, 1030, 64, 3),
                                               ("pull", torch.float32, np.float32, 8192, 512, None),
                                               ("push", torch.float64, np.float64, 4096, 256, 6)):
            steps, omega = 15, 0.7
            sim = SlabSimulation(nx, ny, scene="shear", scheme=scheme, dtype=tdt, omega=omega, device="cuda:0",
                                 p2p_to_self=True, halo=halo)
            assert sim.stream_ordered and sim.overlap
            sim.step(steps)
            vmax = sim.max_velocity()
            want, _ = orc.run("shear", scheme, 8 if nx % 8 == 0 else nx, ny, steps, omega, ndt)
            got = sim.interior()
            if want.shape[2] == nx:
                same = np.array_equal(got.cpu().numpy().view(np.uint8), want.view(np.uint8))
            else:
                col = torch.from_numpy(np.ascontiguousarray(want[:, :, :1])).cuda()
                bits = torch.int32 if tdt == torch.float32 else torch.int64
                same = bool((got.view(bits) == col.view(bits)).all())
            ok = ok and same and vmax == float(orc.max_velocity(want))
        with open(os.path.join(out_dir, "result.txt"), "w") as fh:
            fh.write("ok" if ok else "mismatch")
    finally:
        dist.destroy_process_group()


def test_rccl_transport_with_self_as_both_neighbours():
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker_rccl_self, args=(1, 29541, tmp), nprocs=1, join=True)
        assert open(os.path.join(tmp, "result.txt")).read() == "ok"


def test_two_ranks_on_one_gpu_reproduce_the_oracle(ranks_share_one_gpu):
    from oracle import oracle as orc
    with tempfile.TemporaryDirectory() as tmp:
        store, out = os.path.join(tmp, "store"), os.path.join(tmp, "out.pt")
        mp.spawn(_worker, args=(2, store, out), nprocs=2, join=True)
        results = torch.load(out, weights_only=False)
    for case in CASES:
        scene, scheme, tdt, nx, ny, steps, omega = case
        ndt = np.float32 if tdt == "float32" else np.float64
        want, _ = orc.run(scene, scheme, nx, ny, steps, omega, ndt)
        for halo in (1, 4, 6):
            got, vmax = results[case + (halo,)]
            assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (case, halo)
            assert vmax == float(orc.max_velocity(want)), (case, halo)
