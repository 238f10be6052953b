"""Times the 2-D Cartesian path (lbm_amd.cart) on one rank -- the reference's scheme with a halo ring of one
node, and blocks of several steps on deep frames -- next to the plain single-device loop."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import Simulation, ops
from lbm_amd.cart import CartSimulation
for arith in ("unrolled", "contracted"):
    ops.set_arithmetic(arith)
    for n in (2048, 8192, 16384):
        for dtype in (torch.float32, torch.float64):
            steps = 48
            for name, make in (("cart 1x1, halo ring of one node", lambda: CartSimulation(n, n, dtype=dtype, halo=1)),
                               ("cart 1x1, deep frames", lambda: CartSimulation(n, n, dtype=dtype)),
                               ("cart 1x1, deep frames, push", lambda: CartSimulation(n, n, dtype=dtype, scheme="push")),
                               ("single device, multi-step launches", lambda: Simulation(n, n, dtype=dtype))):
                sim = make()
                sim.step(16)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sim.step(steps)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                depth = getattr(sim, "G", "")
                print(f"{arith:10s} {n:6d}^2 {str(dtype):14s} {name:36s} {str(depth):2s} {dt / steps * 1e3:8.3f} ms/step "
                      f"{n * n * steps / dt / 1e6:9.0f} MLUPS", flush=True)
                del sim
