import os, sys, tempfile
import numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def worker(rank, world, store, cases):
    sys.path.insert(0, ROOT)
    from lbm_amd.slab import SlabSimulation
    from oracle import oracle as orc
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=world)
    for (scheme, nx, ny, steps, overlap) in cases:
        sim = SlabSimulation(nx, ny, scene="shear", scheme=scheme, dtype=torch.float32, omega=0.5, device="cuda:0", overlap=overlap)
        sim.step(steps)
        vmax = sim.max_velocity()
        want, _ = orc.run("shear", scheme, 8, ny, steps, 0.5, np.float32)
        mine = sim.interior()
        col = torch.from_numpy(np.ascontiguousarray(want[:, sim.y_below:sim.y_below + sim.n, :1])).cuda()
        bad = (mine.view(torch.int32) != col.view(torch.int32))
        nbad = int(bad.sum())
        rows = torch.nonzero(bad.any(dim=2).any(dim=0)).flatten().tolist()[:10]
        print(f"rank {rank} {scheme} {nx}x{ny} steps {steps} overlap {overlap}: bad {nbad} rows {rows} vmax {vmax}", flush=True)
        dist.barrier()
    dist.destroy_process_group()

if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    cases = [("push", 16384, 2048, 12, True), ("push", 16384, 2048, 12, False), ("pull", 16384, 2048, 12, True), ("push", 1024, 64, 12, True)]
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(worker, args=(world, os.path.join(tmp, "s"), cases), nprocs=world, join=True)
