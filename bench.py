#!/usr/bin/env python3
"""Headline benchmark: D2Q9 BGK fused collide+stream sweep, MLUPS and fraction of the HBM roofline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision fp32|fp64] [--scheme pull|push]
                    [--scene shear|lid] [--nx NX] [--ny-per-gpu NY]

One "step" is one time step: every node of the lattice relaxed and streamed once.  The time loop
hands runs of up to eight steps to one fused launch (same populations bit for bit, one
pass over HBM instead of six); --max-fuse 1 times the one-step sweep instead.  N=1 runs the single-GPU
configuration (shear-wave decay, 16384 x 16384, fp32, pull streaming: BASELINE configs[2]).  N>1
(launched by torch.distributed.run, one rank per GPU) runs BASELINE configs[4]: every rank owns a
32768 x 4096 y-slab of a 32768 x (4096*N) lattice (weak scaling; --strong: a 32768 x 32768 lattice
split over the ranks) and refreshes up to eight ghost rows per slab edge once per as many steps over
RCCL, overlapped with the interior update; value = all ranks' lattice updates per second.

Inputs are resident in HBM when the clock starts (allocation and initialisation are outside the
timed region); the timed region is bracketed by barrier + device synchronise on both sides and the
maximum over ranks is reported.

The JSON line also carries
  roofline:     the roof that binds the dominant sweep launch, whose duration is measured with HIP
                events on the launch stream.  Two floors are priced at the nominal peaks: `hbm` = the
                bytes the launch moves through HBM (PMC `traffic` = FETCH_SIZE x 2 + WRITE_SIZE when a
                profile of this workload is committed, else its compulsory bytes
                `min_bytes_per_launch`) / 8 TB/s, and `valu` = its vector-ALU wave-instructions (PMC
                SQ_INSTS_VALU of the same profile) x 4 cycles / (1024 SIMDs x 2.4 GHz).  `bound` names
                the larger floor, `achieved` / `peak` are that resource's rate and peak, `frac` =
                achieved / peak = floor / measured time (<= 1).  The algorithmic figure of a one-step
                sweep (72 B per node and time step fp32, 144 B fp64, times the steps the launch
                advances) is kept as `algorithmic_achieved` / `algorithmic_over_peak`: a fused launch
                moves fewer bytes than that count, so it exceeds 1 and bounds nothing.  The same profile also
                records what the part delivers (copy ceiling 6.2 TB/s, engine clock under this load
                ~1.8-2.0 GHz): `hbm_frac_of_copy_ceiling`, `valu.frac_at_profiled_clock`, and the larger
                of those two floors as `binding_as_delivered` / `frac_of_binding_roof_as_delivered`;
  parity:       after the timed region every rank compares its rows with the CPU oracle's run of the
                same global lattice (8 columns wide: the shear wave is x-invariant, so every column
                of the GPU state must equal the oracle's column bit for bit);
  cpu_baseline: the CPU restatement of the reference algorithm (oracle/, OpenMP over rows) timed on
                this host's cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, GB/s
COPY_CEILING_GBS = 6200.0  # best 1:1 read/write stream measured on this part (profiles/r01/copy_ceiling.txt)
N_SIMDS = 256 * 4               # compute units x SIMDs
CLOCK_HZ = 2.4e9                # peak engine clock
VALU_CYCLES_PER_INST = 4        # issue cycles of one (packed) vector-ALU wave-instruction on one SIMD


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--precision", choices=["fp32", "fp64"], default="fp32")
    ap.add_argument("--scheme", choices=["pull", "push"], default="pull")
    ap.add_argument("--scene", choices=["shear", "lid"], default="shear")
    ap.add_argument("--nx", type=int, default=None,
                    help="row length (default: 16384 on one GPU, 32768 on several = BASELINE configs[2] / configs[4])")
    ap.add_argument("--ny-per-gpu", type=int, default=None,
                    help="rows per rank (default: 16384 on one GPU; 4096 on several: the weak-scaling slab of "
                         "BASELINE configs[4], 32768 x 4096 per GPU)")
    ap.add_argument("--ny-global", type=int, default=0,
                    help="strong scaling: fixed global height split over the ranks (overrides --ny-per-gpu), "
                         "e.g. --nx 32768 --ny-global 32768")
    ap.add_argument("--strong", action="store_true",
                    help="several GPUs: the strong-scaling shape of BASELINE configs[4], a 32768 x 32768 lattice split "
                         "over the ranks (= --nx 32768 --ny-global 32768)")
    ap.add_argument("--omega", type=float, default=0.5)
    ap.add_argument("--rolled", action="store_true",
                    help="the reference's UNROLL_LOOPS=false arithmetic in the periodic sweeps (shear scene)")
    ap.add_argument("--arithmetic", choices=["unrolled", "strict", "contracted"], default=None,
                    help="contracted (the default of the library, of ./main and of this script) = the relaxation with the "
                         "multiply-adds fused that the reference's own build flags let a compiler fuse "
                         "(LBM_ARITH_CONTRACTED), checked bit for bit against the oracle's restatement of the same "
                         "contraction and within 1e-5 / 1e-12 of the strict one; strict = unrolled = every operation "
                         "rounded where the reference's text puts it (LBM_ARITHMETIC=strict)")
    ap.add_argument("--layout", choices=["soa", "aos"], default="soa",
                    help="aos = [NY][NX][9] arrays, the reference's COALESCING=false (single GPU only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the comparison with the CPU oracle after the run")
    ap.add_argument("--cpu-rows", type=int, default=2048, help="rows of the CPU-baseline sample")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of the CPU-baseline sample")
    ap.add_argument("--no-overlap", action="store_true", help="exchange halos on the compute stream")
    ap.add_argument("--slab", action="store_true",
                    help="N=1 only: run the y-slab code path with this GPU as its own periodic neighbour "
                         "(halo rows through RCCL send/recv to self) to price the decomposition overhead")
    ap.add_argument("--max-fuse", type=int, default=0,
                    help="upper bound on time steps per launch (0 = library default of 7 or 8, 1 = one-step sweeps)")
    ap.add_argument("--threads-x", type=int, default=0, help="workgroup width hint (0 = built-in)")
    ap.add_argument("--waves-per-cu", type=int, default=0, help="resident-wave cap hint (0 = built-in)")
    return ap.parse_args()


def cpu_baseline(nx, rows, budget_s, precision, scheme, scene, omega):
    """Time the CPU oracle (the checker, here only as a reported baseline) on a bounded sample of
    the same workload: same row length, `rows` rows, as many steps as fit in `budget_s` seconds."""
    import numpy as np
    from oracle import oracle as orc

    dt = np.float32 if precision == "fp32" else np.float64
    typ = np.dtype(dt).type
    f, buf = orc.alloc(nx, rows, dt), orc.alloc(nx, rows, dt)
    om, rho, u = typ(omega), typ(1.0), typ(0.1)
    if scene == "lid":
        orc.init_rest(f, rho)
        sweep = (lambda a, b: orc.pull_lid_driven(a, b, om, rho, u)) if scheme == "pull" else \
                (lambda a, b: orc.push_lid_driven(a, b, om, rho, u))
    else:
        orc.init_shearwave(f, rho, u)
        sweep = (lambda a, b: orc.pull_periodic_no_mpi(a, b, om)) if scheme == "pull" else \
                (lambda a, b: orc.push_periodic_no_mpi(a, b, om))
    sweep(f, buf)  # first touch of buf outside the timed part
    f, buf = buf, f
    steps, t0 = 0, time.perf_counter()
    while True:
        sweep(f, buf)
        f, buf = buf, f
        steps += 1
        dt_s = time.perf_counter() - t0
        if dt_s >= budget_s or steps >= 100000:
            break
    return {
        "value": nx * rows * steps / dt_s / 1e6,
        "unit": "MLUPS",
        "cores": orc.num_threads(),
        "kind": "port",
        "sample": f"{scene} {scheme} {precision} {nx}x{rows} slab of the workload, {steps} steps in {dt_s:.1f} s, "
                  "C restatement of the reference algorithm (oracle/), OpenMP over rows; the reference's own "
                  "Kokkos build is not buildable offline",
    }


def pmc_profile(workload_key):
    """Per-launch counters measured with rocprofv3 PMC passes (HBM bytes, vector-ALU instructions),
    if a summary for exactly this workload and step count per launch has been committed under
    profiles/ (bench.py cannot collect counters itself)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh).get(workload_key, {})
    except (OSError, ValueError):
        return {}


def check_against_oracle(sim, scene, scheme, precision, nx, ny_glob, steps_total, omega, y_below, rows,
                         arithmetic="unrolled"):
    """The checker: rows [y_below, y_below + rows) of the oracle's run of the global lattice, 8
    columns wide, against every column of this rank's rows.  Returns (checked, equal, max_abs):
    bit equality with the oracle in the same arithmetic, and the largest absolute difference of any
    population from the oracle in the strict arithmetic (0.0 when that is the arithmetic of the run)."""
    if scene != "shear":
        return False, None, None     # the cavity is not x-invariant; covered by the test-suite instead
    import numpy as np
    import torch
    from oracle import oracle as orc
    ndt = np.float32 if precision == "fp32" else np.float64
    with orc.arithmetic(arithmetic):
        want, _ = orc.run("shear", scheme, 8, ny_glob, steps_total, omega, ndt)
    col = torch.from_numpy(np.ascontiguousarray(want[:, y_below:y_below + rows, :1])).to(sim.f.device)
    mine = sim.interior() if hasattr(sim, "interior") else sim.f
    if mine.dim() == 3 and mine.shape[0] != 9:          # [NY, NX, 9] arrays
        mine = mine.permute(2, 0, 1)
    bits = torch.int32 if precision == "fp32" else torch.int64
    equal = all(bool((mine[q].contiguous().view(bits) == col[q].view(bits)).all()) for q in range(9))
    max_abs = 0.0
    if arithmetic != "unrolled":
        strict, _ = orc.run("shear", scheme, 8, ny_glob, steps_total, omega, ndt)
        scol = torch.from_numpy(np.ascontiguousarray(strict[:, y_below:y_below + rows, :1])).to(sim.f.device)
        max_abs = max(float((mine[q].double() - scol[q].double()).abs().max()) for q in range(9))
    return True, equal, max_abs


def main():
    args = parse()
    # Libraries underneath (RCCL prints a version banner when its first communicator comes up) write to
    # file descriptor 1; the one JSON line must be the only thing on stdout, so everything before it
    # goes to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device")
    # LBM_BENCH_REHEARSE=1: rehearsal of the multi-rank code path on a box with fewer GPUs than
    # ranks (all ranks share the visible GPUs, transport gloo); never a measurement
    rehearse = os.environ.get("LBM_BENCH_REHEARSE") == "1"
    dev_index = local_rank % torch.cuda.device_count() if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    if args.max_fuse:
        os.environ["LBM_FUSED_MAX_STEPS"] = str(args.max_fuse)   # read by lbm_plan_steps at every call
    from lbm_amd import Simulation, SlabSimulation, capi, ops
    if args.threads_x or args.waves_per_cu:
        capi.call("lbm_set_launch_shape", args.threads_x, args.waves_per_cu)

    dtype = torch.float32 if args.precision == "fp32" else torch.float64
    bytes_per_node = 72 if args.precision == "fp32" else 144
    # Shapes nobody asked for otherwise are BASELINE.json's: configs[2] on one GPU (16384 x 16384), configs[4]
    # on several (weak: a 32768 x 4096 slab per GPU; --strong: 32768 x 32768 split over the ranks)
    baseline_shape = args.nx is None and args.ny_per_gpu is None and args.ny_global == 0
    if args.strong:
        if args.ny_global or args.ny_per_gpu:
            raise SystemExit("--strong fixes the global height; do not combine it with --ny-global / --ny-per-gpu")
        args.ny_global = 32768
        args.nx = args.nx or 32768
    nx = args.nx or (16384 if world == 1 else 32768)
    ny_loc = args.ny_per_gpu or (16384 if world == 1 else 4096)
    ny_glob = ny_loc * world
    strong = args.ny_global > 0
    if strong:
        ny_glob = args.ny_global
        ny_loc = -(-ny_glob // world)   # the largest slab (remainder rows go to the lowest ranks)

    if args.arithmetic == "strict":
        args.arithmetic = "unrolled"
    if args.arithmetic is None:
        args.arithmetic = "unrolled" if (args.rolled or args.layout == "aos") else "contracted"
    if args.rolled:
        ops.set_unroll_loops(False)
    if args.arithmetic == "contracted" and args.rolled:
        raise SystemExit("--arithmetic contracted does not go with --rolled")
    if not args.rolled:
        ops.set_arithmetic(args.arithmetic)   # whatever LBM_ARITHMETIC says: the line names what ran
    if args.layout == "aos":
        if world != 1 or args.slab:
            raise SystemExit("--layout aos runs on one GPU without the slab path")
        ops.set_layout("aos")
    if world == 1 and args.slab:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        sim = SlabSimulation(nx, ny_glob, scene=args.scene, scheme=args.scheme, dtype=dtype, omega=args.omega,
                             device=device, overlap=not args.no_overlap, p2p_to_self=True)
    elif world == 1:
        sim = Simulation(nx, ny_loc, scene=args.scene, scheme=args.scheme, dtype=dtype,
                         omega=args.omega, device=device)
    else:
        sim = SlabSimulation(nx, ny_glob, scene=args.scene, scheme=args.scheme, dtype=dtype,
                             omega=args.omega, device=device, overlap=not args.no_overlap)

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    # The launch plan of the timed run: one entry per sweep launch (single device) or block (slabs).
    def plan_of(n):
        probe_rows = ny_loc
        out, kmax = [], (sim.G if hasattr(sim, "G") else None)
        while n > 0:
            if kmax is None:
                k = ops.plan_steps(sim.f, probe_rows, n, lid=args.scene == "lid")
            else:
                k = -(-n // -(-n // kmax))
            out.append(k)
            n -= k
        return out

    plan = plan_of(args.steps)
    # Warm-up: the W steps asked for, then one launch of every step count of the timed plan that the
    # warm-up did not already run (the first launch of a kernel pays for loading its code).
    sim.step(args.warmup)
    warmed = set(sim.launch_steps)
    extra_warmup = 0
    for k in sorted(set(plan) - warmed):
        sim.step(k)
        extra_warmup += k
    stream = torch.cuda.current_stream(device)  # the stream the sweeps are launched on
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(len(plan) + 1)]
    sync()
    t0 = time.perf_counter()
    marks[0].record(stream)
    for i, k in enumerate(plan):
        sim.step(k)
        marks[i + 1].record(stream)
    sync()
    elapsed = time.perf_counter() - t0
    assert sim.launch_steps[-len(plan):] == plan, (sim.launch_steps[-len(plan):], plan)
    launch_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(plan))]
    steps_total = args.warmup + extra_warmup + args.steps

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the run must still be a physical flow (no NaN, right magnitude)
    vmax = sim.max_velocity()
    if not (0.0 <= vmax < 1.0):
        raise SystemExit(f"benchmark state is not physical: max |u| = {vmax}")

    # parity: every rank checks its own rows against the oracle's run of the global lattice
    if hasattr(sim, "y_below"):
        my_below, my_rows = sim.y_below, sim.n
    else:
        my_below, my_rows = 0, ny_loc
    checked, equal, max_abs = check_against_oracle(sim, args.scene, args.scheme, args.precision, nx, ny_glob,
                                                   steps_total, args.omega, my_below, my_rows, args.arithmetic) \
        if (not args.rolled and not args.no_parity) else (False, None, None)
    budget = 1e-5 if args.precision == "fp32" else 1e-12
    report = {"rank": rank, "device": f"cuda:{dev_index}", "name": torch.cuda.get_device_name(dev_index),
              "rows": [my_below, my_below + my_rows], "parity_checked": checked, "bit_identical": equal,
              "max_abs_vs_strict_oracle": max_abs}
    if world > 1:
        reports = [None] * world
        dist.all_gather_object(reports, report)
    else:
        reports = [report]
    if any(r["parity_checked"] and (not r["bit_identical"] or r["max_abs_vs_strict_oracle"] > budget) for r in reports):
        raise SystemExit(f"benchmark state differs from the CPU oracle: {reports}")

    # N=1: the same run in the strict arithmetic (the library's default), reported beside the headline
    strict_leg = None
    if world == 1 and not args.slab and args.arithmetic == "contracted" and not args.no_parity:
        del sim
        torch.cuda.empty_cache()
        ops.set_arithmetic("unrolled")
        sim2 = Simulation(nx, ny_loc, scene=args.scene, scheme=args.scheme, dtype=dtype, omega=args.omega, device=device)
        plan2, n = [], args.steps
        while n > 0:
            plan2.append(ops.plan_steps(sim2.f, ny_loc, n, lid=args.scene == "lid"))
            n -= plan2[-1]
        sim2.step(args.warmup)
        for k in sorted(set(plan2) - set(sim2.launch_steps)):
            sim2.step(k)
        warm2 = sim2.t
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for k in plan2:
            sim2.step(k)
        torch.cuda.synchronize(device)
        el2 = time.perf_counter() - t0
        c2, e2, _ = check_against_oracle(sim2, args.scene, args.scheme, args.precision, nx, ny_glob, warm2 + args.steps,
                                         args.omega, 0, ny_loc, "unrolled")
        strict_leg = {"value": nx * ny_loc * args.steps / el2 / 1e6, "unit": "MLUPS", "launch_plan": plan2,
                      "arithmetic": "unrolled (every operation rounded where the reference's text puts it)",
                      "parity_checked": c2, "bit_identical": e2}
        if c2 and not e2:
            raise SystemExit("strict-arithmetic leg differs from the CPU oracle")
        ops.set_arithmetic("contracted")
        sim = sim2

    if rank == 0:
        updates = nx * ny_glob * args.steps
        mlups = updates / elapsed / 1e6
        # dominant kernel = the (interior) sweep at the step count most launches of the plan take
        n_launch = len(plan)
        k_dom = max(set(plan), key=lambda k: (plan.count(k), k))
        dom_ms = [ms for ms, k in zip(launch_ms, plan) if k == k_dom]
        launch_s = sum(dom_ms) / len(dom_ms) * 1e-3
        steps_per_launch = args.steps / n_launch
        alg_bytes = nx * ny_loc * bytes_per_node * k_dom
        min_bytes = nx * ny_loc * bytes_per_node
        achieved = alg_bytes / launch_s / 1e9
        workload = (f"{'shear-wave decay' if args.scene == 'shear' else 'lid-driven cavity'} "
                    f"{nx}x{ny_glob} {args.precision} {args.scheme}")
        if world > 1:
            workload += f", {world} y-slabs of {nx}x{ny_loc}"
        if baseline_shape or args.strong:
            if world == 1:
                if (args.scene, args.scheme, args.precision, args.layout, args.rolled) == ("shear", "pull", "fp32", "soa", False):
                    workload += " (BASELINE configs[2])"
            else:
                workload += (" (BASELINE configs[4], strong scaling: 32768x32768 over the GPUs)" if strong else
                             " (BASELINE configs[4], weak scaling: a 32768x4096 slab per GPU)")
        key = f"{args.scene}-{args.scheme}-{args.precision}-{nx}x{ny_loc}" + ("-aos" if args.layout == "aos" else "") + ("-rolled" if args.rolled else "") + ("-contracted" if args.arithmetic == "contracted" else "")
        # profiles are keyed by the step count of the dominant launches
        if k_dom != 1:
            key += f"-fused{k_dom}"
        prof = pmc_profile(key) if world == 1 and not args.slab else {}
        traffic = prof.get("hbm_bytes_per_launch")
        valu_insts = prof.get("valu_insts_per_launch")
        hbm_floor_s = (traffic if traffic else min_bytes) / (HBM_PEAK_GBS * 1e9)
        valu_floor_s = valu_insts * VALU_CYCLES_PER_INST / (N_SIMDS * CLOCK_HZ) if valu_insts else None
        # the engine clock the profiled launches actually ran at (GRBM_GUI_ACTIVE / duration): power-capped
        # well below the 2.4 GHz peak in these kernels, so the same instructions price a higher floor
        clock_seen = prof.get("gfx_clock_ghz_during_launch")
        valu_floor_seen_s = (valu_insts * VALU_CYCLES_PER_INST / (N_SIMDS * clock_seen * 1e9)
                             if valu_insts and clock_seen else None)
        floors = {"hbm": hbm_floor_s}
        if valu_floor_s:
            floors["valu"] = valu_floor_s
        binding = max(floors, key=floors.get)
        # the same two floors at what the part delivers: copy ceiling for the bytes, the profiled clock for the issue
        delivered = {"hbm": hbm_floor_s * HBM_PEAK_GBS / COPY_CEILING_GBS}
        if valu_floor_seen_s:
            delivered["valu"] = valu_floor_seen_s
        binding_delivered = max(delivered, key=delivered.get)
        # the roof that binds the dominant launch, at the nominal peaks: HBM bytes moved (counters of a committed
        # profile, else the compulsory bytes) against 8 TB/s, or vector-ALU wave-instructions against
        # 1024 SIMDs x 2.4 GHz / 4 cycles -- whichever floor is the larger
        hbm_bytes = traffic if traffic else min_bytes
        valu_peak_ginst = N_SIMDS * CLOCK_HZ / VALU_CYCLES_PER_INST / 1e9
        if binding == "valu":
            roof_achieved, roof_peak, roof_unit = valu_insts / launch_s / 1e9, valu_peak_ginst, "Gwave-inst/s"
        else:
            roof_achieved, roof_peak, roof_unit = hbm_bytes / launch_s / 1e9, HBM_PEAK_GBS, "GB/s"
        out = {
            "metric": "MLUPS",
            "value": mlups,
            "unit": "MLUPS",
            "arithmetic": "rolled loops (UNROLL_LOOPS=false)" if args.rolled else
                          {"contracted": "contracted (library default)", "unrolled": "strict (LBM_ARITHMETIC=strict)"}[args.arithmetic],
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "f64",
            "data": "synthetic" if not rehearse else "synthetic (REHEARSAL over gloo on shared GPUs: not a measurement)",
            "config": {
                "workload": workload,
                "scene": args.scene, "scheme": args.scheme, "layout": args.layout,
                "arithmetic": "rolled loops (UNROLL_LOOPS=false)" if args.rolled else args.arithmetic, "nx": nx, "ny_global": ny_glob,
                "ny_per_gpu": ny_loc, "omega": args.omega,
                "decomposition": ("single device" if world == 1 and not args.slab else
                                  f"{world} y-slab(s), {sim.G} ghost row(s) per edge refreshed every "
                                  f"{sim.G} step(s), "
                                  + ("RCCL" if (dist.is_initialized() and dist.get_backend() == "nccl") else "gloo (rehearsal)")
                                  + " send/recv"),
                "steps_per_launch": steps_per_launch,
                "launch_plan": plan,
                "warmup_steps_run": args.warmup + extra_warmup,
                "roofline_mlups_per_gpu": HBM_PEAK_GBS * 1e3 / bytes_per_node,
                "frac_of_roofline_whole_job": mlups / (world * HBM_PEAK_GBS * 1e3 / bytes_per_node),
                "max_velocity_after_run": vmax,
            },
            "roofline": {
                "bound": binding,
                "achieved": roof_achieved,
                "peak": roof_peak,
                "unit": roof_unit,
                "frac": roof_achieved / roof_peak,
                "traffic": traffic,
                "algorithmic_achieved": achieved,
                "algorithmic_over_peak": achieved / HBM_PEAK_GBS,
                "kernel": (f"{args.scheme} {'periodic' if args.scene == 'shear' else 'lid-driven'} sweep, "
                           f"{k_dom} time step(s) per launch"),
                "steps_per_launch": k_dom,
                "launches_timed": len(dom_ms),
                "algorithmic_bytes_per_launch": alg_bytes,
                "min_bytes_per_launch": min_bytes,
                "hbm_frac": hbm_floor_s / launch_s,
                "hbm_bytes_source": "rocprofv3 FETCH_SIZE x2 + WRITE_SIZE of a committed profile" if traffic
                                    else "compulsory bytes (no profile of this workload committed)",
                "valu": ({"insts_per_launch": valu_insts, "cycles_per_inst": VALU_CYCLES_PER_INST,
                          "simds": N_SIMDS, "clock_ghz": CLOCK_HZ / 1e9, "floor_ms": valu_floor_s * 1e3,
                          "frac": valu_floor_s / launch_s,
                          "clock_ghz_profiled": clock_seen,
                          "floor_ms_at_profiled_clock": valu_floor_seen_s * 1e3 if valu_floor_seen_s else None,
                          "frac_at_profiled_clock": valu_floor_seen_s / launch_s if valu_floor_seen_s else None}
                         if valu_floor_s else None),
                "hbm_frac_of_copy_ceiling": hbm_floor_s * HBM_PEAK_GBS / COPY_CEILING_GBS / launch_s,
                "valu_frac": (valu_floor_s / launch_s) if valu_floor_s else None,
                "binding": binding,
                "frac_of_binding_roof": floors[binding] / launch_s,
                "binding_as_delivered": binding_delivered,
                "frac_of_binding_roof_as_delivered": delivered[binding_delivered] / launch_s,
                "profile": prof.get("source"),
                "launch_ms": launch_s * 1e3,
                "note": ("frac = the larger of the launch's two floors at the nominal peaks (HBM bytes actually moved / "
                         "8 TB/s; vector-ALU wave-instructions x 4 cycles / 1024 SIMDs / 2.4 GHz) over its measured "
                         "duration; one launch advances several time steps per pass over HBM, so the algorithmic rate "
                         "(72/144 B per node and step: algorithmic_achieved) can exceed the HBM peak and bounds nothing")
                        if k_dom != 1 else "one time step per launch",
            },
        }
        out["parity_checked"] = all(r["parity_checked"] for r in reports)
        if strict_leg:
            out["strict_arithmetic"] = strict_leg
        out["parity"] = {"against": ("CPU oracle in the same arithmetic, 8-column run of the global lattice, every column "
                                     "bit for bit; max_abs_vs_strict_oracle = largest difference of any population from "
                                     f"the oracle in the strict arithmetic (budget {budget:g})"),
                         "steps_compared": steps_total, "ranks": reports}
        out["world"] = {"size": world, "backend": (dist.get_backend() if dist.is_initialized() else None),
                        "transport": ("RCCL" if dist.is_initialized() and dist.get_backend() == "nccl"
                                      else (dist.get_backend() if dist.is_initialized() else "none"))}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nx, args.cpu_rows, args.cpu_seconds, args.precision,
                                               args.scheme, args.scene, args.omega)
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)

    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
