"""lbm_amd -- MI355X-native D2Q9 BGK lattice-Boltzmann hot path.

Layout:
  csrc/          hand-written gfx950 HIP kernels and the C ABI (include/lbm_hip.h)
  capi.py        ctypes binding of the C ABI (fails loudly when the library is missing)
  ops.py         the reference's operator functions (init.h, pull-push.h, io.h) on device tensors
  sim.py         the reference's single-device time loop (main.cpp:31-89) and its text outputs
  slab.py        y-slab decomposition over ranks with halo-row exchange (torch.distributed / RCCL)
  cart.py        the reference's 2-D Cartesian decomposition (halo ring, inner / outer sweeps)

PyTorch is used only for device memory, streams and torch.distributed; all lattice arithmetic runs
in the HIP kernels.  There is no CPU fallback.
"""
from . import capi, ops  # noqa: F401
from .sim import Simulation  # noqa: F401
from .slab import SlabSimulation  # noqa: F401
from .cart import CartSimulation  # noqa: F401

__all__ = ["capi", "ops", "Simulation", "SlabSimulation", "CartSimulation"]
