"""Times one-step sweeps against the fused kernels asked for one step (rows 0..ny-2 so that the call is
not forwarded to the one-step kernels)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lbm_amd import ops
nx = ny = 16384
for dtype in (torch.float32, torch.float64):
    f = torch.full((9, ny, nx), 1.0 / 9, dtype=dtype, device="cuda")
    b = torch.empty_like(f)
    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    cases = {
        "pull periodic one-step": lambda: ops.pull_periodic_no_mpi(f, b, nx, ny, 0.5),
        "pull periodic fused k=1": lambda: ops.pull_periodic_fused(f, b, 0.5, 0, ny - 1, True, 1),
        "push periodic one-step": lambda: ops.push_periodic_no_mpi(f, b, nx, ny, 0.5),
        "push periodic fused k=1": lambda: ops.push_periodic_fused(f, b, 0.5, 0, ny - 1, True, 1),
        "push lid one-step": lambda: ops.push_lid_driven(f, b, nx, ny, 0.5, 1.0, 0.1),
        "push lid fused k=1": lambda: ops.push_lid_fused(f, b, 0.5, 1.0, 0.1, 0, ny - 1, 0, ny, 1),
        "pull lid one-step": lambda: ops.pull_lid_driven(f, b, nx, ny, 0.5, 1.0, 0.1),
        "pull lid fused k=1": lambda: ops.pull_lid_fused(f, b, 0.5, 1.0, 0.1, 0, ny - 1, 0, ny, 1),
    }
    for name, fn in cases.items():
        ms = timeit(fn)
        print(f"{str(dtype):14s} {name:26s} {ms:7.3f} ms  {nx * ny / ms / 1e3:9.0f} MLUPS", flush=True)
    del f, b
