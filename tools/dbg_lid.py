import numpy as np, torch, sys
sys.path.insert(0, '.')
from oracle import oracle as orc
from lbm_amd import ops
rng = np.random.default_rng(3)
for (nx, ny, dt) in ((16, 5, np.float32), (8, 6, np.float64)):
    f = (1.0/9 + 0.02*rng.standard_normal((9, ny, nx))).astype(dt)
    want = np.full_like(f, np.nan); orc.push_lid_driven(f, want, dt(1.3), dt(1.0), dt(0.1))
    d_f = torch.from_numpy(f).cuda(); d_b = torch.full_like(d_f, float('nan'))
    ops.push_lid_driven(d_f, d_b, nx, ny, 1.3, 1.0, 0.1)
    got = d_b.cpu().numpy()
    bad = ~((got == want) | (np.isnan(got) & np.isnan(want)))
    print(nx, ny, dt.__name__, "bad:", bad.sum())
    for q, y, x in np.argwhere(bad)[:40]:
        print("  q", q, "y", y, "x", x, got[q, y, x], want[q, y, x])
