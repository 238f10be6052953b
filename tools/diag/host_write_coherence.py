"""Diagnostic (GPU box): does a plain host store into hipMalloc'ed memory (what a transport that knows
nothing about streams does when it is handed a device pointer) reach a later kernel?
A kernel reads a buffer (its lines now sit in the XCDs' L2 caches), the host overwrites the buffer
through its CPU mapping with ctypes.memmove, the device is synchronised, a second kernel reads it."""
import ctypes
import sys

import numpy as np
import torch

n = 1 << 20
x = torch.zeros(n, dtype=torch.float32, device="cuda")
x.fill_(1.0)
torch.cuda.synchronize()
before = float(x.sum())                      # pulls the lines into L2
host = np.full(n, 2.0, dtype=np.float32)
print("device pointer", hex(x.data_ptr()), "sum before", before, flush=True)
ctypes.memmove(ctypes.c_void_p(x.data_ptr()), host.ctypes.data_as(ctypes.c_void_p), host.nbytes)
torch.cuda.synchronize()
after = x.clone()                            # a kernel reading x
stale = int((after != 2.0).sum())
print(f"after the host store: {stale} of {n} values still old (sum {float(after.sum())})", flush=True)
y = torch.from_numpy(host).cuda()            # the staged path: runtime copy
print("staged copy equals host data:", bool((y == 2.0).all()), flush=True)
