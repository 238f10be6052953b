"""Times the sweeps on [NY][NX][9] arrays (layout "aos") next to the [9][NY][NX] ones, 16384^2."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import ops
nx = ny = int(sys.argv[1]) if len(sys.argv) > 1 else 16384


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for dtype in (torch.float32, torch.float64):
    for lay in ("soa", "aos"):
        with ops.layout(lay):
            f = torch.full(ops.population_shape(nx, ny), 1.0 / 9, dtype=dtype, device="cuda")
            b = torch.empty_like(f)
            for k in (1, 4, 6):
                cases = {
                    "pull periodic": lambda: ops.pull_periodic_fused(f, b, 0.5, 0, ny, True, k),
                    "push periodic": lambda: ops.push_periodic_fused(f, b, 0.5, 0, ny, True, k),
                    "pull lid": lambda: ops.pull_lid_fused(f, b, 0.5, 1.0, 0.1, 0, ny, 0, ny, k),
                    "push lid": lambda: ops.push_lid_fused(f, b, 0.5, 1.0, 0.1, 0, ny, 0, ny, k),
                }
                for name, fn in cases.items():
                    ms = timeit(fn)
                    print(f"{str(dtype):14s} {lay} k={k} {name:14s} {ms:8.3f} ms  {k * nx * ny / ms / 1e6:8.1f} GLUPS", flush=True)
            del f, b
