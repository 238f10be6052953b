"""Does the distance between f and buf matter?  One pool, the two arrays carved out of it at chosen
offsets, multi-step launches timed with events (16384^2 fp32).  Diagnostic, not product."""
import sys

import torch

sys.path.insert(0, ".")
from lbm_amd import ops  # noqa: E402

nx = ny = 16384
n = 9 * nx * ny
pool = torch.empty(2 * n + (64 << 20), dtype=torch.float32, device="cuda")
print("pool at", hex(pool.data_ptr()))
plain_f = torch.empty((9, ny, nx), dtype=torch.float32, device="cuda")
plain_b = torch.empty_like(plain_f)
print("torch.empty pair at", hex(plain_f.data_ptr()), hex(plain_b.data_ptr()), "distance", (plain_b.data_ptr() - plain_f.data_ptr()) / n / 4)
del plain_f, plain_b


def run(skew_bytes, arith, k, reps=6):
    f = pool[:n].view(9, ny, nx)
    off = n + skew_bytes // 4
    b = pool[off:off + n].view(9, ny, nx)
    with ops.Context(arithmetic=arith):
        ops.init_shearwave(f, 1.0, 0.1, 0, ny)
        for _ in range(2):
            ops.pull_periodic_fused(f, b, 0.5, 0, ny, True, k)
            ops.pull_periodic_fused(b, f, 0.5, 0, ny, True, k)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for _ in range(reps):
            ops.pull_periodic_fused(f, b, 0.5, 0, ny, True, k)
            ops.pull_periodic_fused(b, f, 0.5, 0, ny, True, k)
        e[1].record()
        torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]) / (2 * reps)


for arith, k in (("strict", 7), ("contracted", 7), ("contracted", 8)):
    for skew in (0, 256, 4096, 32 << 10, 64 << 10, 256 << 10, 1 << 20, (1 << 20) + 4096, 2 << 20, 3 << 20, 16 << 20, (16 << 20) + (96 << 10)):
        print(f"{arith:10s} K={k} skew {skew:>9d} B  {run(skew, arith, k):7.3f} ms", flush=True)
