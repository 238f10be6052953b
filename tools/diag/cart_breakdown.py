"""Where a block of the 2-D decomposition on deep frames spends its time on one rank (sweep, exchange, pack / unpack)."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lbm_amd import ops
from lbm_amd.cart import CartSimulation
ops.set_arithmetic("contracted")
sim = CartSimulation(16384, 16384, dtype=torch.float32)
sim.step(16); torch.cuda.synchronize()
def t(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
G = sim.G
print("G", G, "block ms", t(lambda: sim._block_deep(G)))
print("sweep ms", t(lambda: sim.backend.periodic_fused(sim.scheme, sim.f, sim.buf, sim.omega, G, G + sim.ny, G)))
print("exchange ms", t(sim._exchange_deep))
f, nx, ny = sim.f, sim.nx, sim.ny
print("col pack ms", t(lambda: f[:, G:G + ny, nx:nx + G].contiguous()))
print("col unpack ms", t(lambda: f[:, G:G + ny, 0:G].copy_(sim._col[0])))
print("row pack ms", t(lambda: f[:, ny:ny + G, :].contiguous()))
print("row unpack ms", t(lambda: f[:, 0:G, :].copy_(sim._row[0])))
