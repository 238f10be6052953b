#!/usr/bin/env python3
"""Do HIP graphs shorten the gaps between the dependent launches of a small lattice?  1024^2, runs of 500 steps
(100 launches of five) as stream launches and as replays of one captured graph."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lbm_amd import Simulation  # noqa: E402


def main():
    for arith in ("contracted", "strict"):
        for scene, scheme in (("shear", "pull"), ("lid", "push")):
            sim = Simulation(1024, 1024, scene=scene, scheme=scheme, arithmetic=arith)
            sim.step(500)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                sim.step(500)
            torch.cuda.synchronize()
            plain = time.perf_counter() - t0
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                sim.step(500)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    sim.step(500)       # an even number of launches: f and buf are back where they were
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    g.replay()
                torch.cuda.synchronize()
                graph = time.perf_counter() - t0
            n = 1024 * 1024 * 500 * 20 / 1e6
            print(f"{arith:10s} {scene:5s} {scheme:4s}  stream launches {n / plain:9.0f} MLUPS   graph replays {n / graph:9.0f} MLUPS", flush=True)


def lanes():
    """One fp32 node per lane (twice the wavefronts, no packed arithmetic) against two, strict arithmetic."""
    from lbm_amd import ops
    for nx in (512, 1024, 1536):
        for scene, scheme in (("shear", "pull"), ("lid", "push")):
            row = []
            for nodes in (0, 1):
                with ops.tuning("fused_lane_nodes", nodes):
                    sim = Simulation(nx, nx, scene=scene, scheme=scheme, arithmetic="strict")
                    sim.step(500)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    sim.step(5000)
                    torch.cuda.synchronize()
                    row.append(nx * nx * 5000 / (time.perf_counter() - t0) / 1e6)
            print(f"strict {nx}^2 {scene:5s} {scheme:4s}  two nodes per lane {row[0]:9.0f} MLUPS   one node per lane {row[1]:9.0f} MLUPS", flush=True)


if __name__ == "__main__":
    main()
    lanes()
