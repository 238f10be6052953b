"""Times the 2-D Cartesian path (lbm_amd.cart) on one rank next to the plain single-device loop."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import Simulation
from lbm_amd.cart import CartSimulation
for n in (2048, 8192, 16384):
    for dtype in (torch.float32, torch.float64):
        steps = 20
        for name, make in (("cart 1x1 (halo ring, inner + outer)", lambda: CartSimulation(n, n, dtype=dtype)),
                           ("single device, one step per launch", lambda: Simulation(n, n, dtype=dtype))):
            sim = make()
            step = (lambda k: sim.step(k)) if isinstance(sim, CartSimulation) else (lambda k: sim.step(k, max_fuse=1))
            step(3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step(steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"{n:6d}^2 {str(dtype):14s} {name:38s} {dt / steps * 1e3:8.3f} ms/step {n * n * steps / dt / 1e6:9.0f} MLUPS", flush=True)
            del sim
