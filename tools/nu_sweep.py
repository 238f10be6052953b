#!/usr/bin/env python3
"""Headless equivalent of the reference's shear-wave.py and problem-size.py (no plots, no LaTeX):

  * viscosity sweep: for each omega run `-nx N -ny N -of 500 -s STEPS -omv`, fit
    a*exp(-nu*(2*pi/NY)^2*t)+c as shear-wave.py:78-80 does and compare with nu = (1/omega - 1/2)/3
    (shear-wave.py:55);
  * problem-size sweep: MLUPS of `-s 100 -of 0` runs over square lattices (problem-size.py:61-71).

Runs on the HIP device through lbm_amd.Simulation; prints one table each.  Usage:
    python tools/nu_sweep.py [--n 1000] [--steps 200000] [--omegas 0.2 0.6 1.0 1.4 1.8] [--sizes 1024 4096 16384]
"""
import argparse
import os
import sys
import time

import numpy as np
from scipy.optimize import curve_fit

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=200000)   # shear-wave.py:60
    ap.add_argument("--every", type=int, default=500)
    ap.add_argument("--omegas", type=float, nargs="*", default=[0.2, 0.6, 1.0, 1.4, 1.8])
    ap.add_argument("--sizes", type=int, nargs="*", default=[1024, 2048, 4096, 8192, 16384])
    ap.add_argument("--precision", choices=["fp32", "fp64"], default="fp32")
    args = ap.parse_args()
    import torch
    from lbm_amd import Simulation
    dtype = torch.float32 if args.precision == "fp32" else torch.float64

    print(f"viscosity sweep, {args.n}x{args.n} {args.precision}, {args.steps} steps, sample every {args.every}")
    print(f"{'omega':>6} {'nu fitted':>10} {'nu analytic':>12} {'rel. error':>11}")
    for om in args.omegas:
        sim = Simulation(args.n, args.n, scene="shear", scheme="pull", dtype=dtype, omega=om)
        text = sim.run(args.steps, args.every, "omv")
        us = np.array([float(v) for v in text.split(",")[:-1]])
        ts = np.arange(len(us), dtype=np.float64) * args.every
        model = lambda x, a, nu, c: a * np.exp(-nu * (2 * np.pi / args.n) ** 2 * x) + c
        (_, nu, _), _ = curve_fit(model, ts, us, (0.2 ** 2, 1, 0))
        want = (1.0 / om - 0.5) / 3.0
        print(f"{om:6.2f} {nu:10.6f} {want:12.6f} {abs(nu - want) / want:11.2e}")
        del sim

    print(f"\nproblem-size sweep, {args.precision} pull, 100 steps each (time loop only)")
    print(f"{'N':>7} {'MLUPS':>10}")
    for n in args.sizes:
        sim = Simulation(n, n, scene="shear", scheme="pull", dtype=dtype, omega=0.5)
        sim.step(12)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sim.step(100)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{n:7d} {n * n * 100 / dt / 1e6:10.0f}")
        del sim
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
