"""The `lbm` command-line driver (build/main): flag surface, text formats, stderr contract.

What the upstream scripts rely on (reference benchmark.py:30-38, shear-wave.py:60-64,
shearwave-video.py:26-32, lid-driven-stationary.py:41-57): the binary is build/main, run with
cwd=build; results go to stdout or the -f file; the LAST stderr line carries the nanoseconds
between the last two '|'."""
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "build")
MAIN = os.path.join(BUILD, "main")


def run_main(*args, check=True, exe="main"):
    if not os.path.exists(os.path.join(BUILD, exe)):
        pytest.fail(f"build/{exe} is missing: run `python __graft_entry__.py` (cmake target `main`) first")
    p = subprocess.run([f"./{exe}", *map(str, args)], cwd=BUILD, capture_output=True, text=True, timeout=600)
    if check:
        assert p.returncode == 0, p.stderr
    return p


# ---- no GPU needed: parsing happens before any device call -----------------------------------
def test_version_and_help():
    assert run_main("--version").stdout.strip() == "1.0"
    assert run_main("-v").stdout.strip() == "1.0"
    h = run_main("--help").stdout
    for flag in ("-nx", "-ny", "-of", "--output-frequency", "-s", "--steps", "-w", "--omega", "-rho", "--density",
                 "-u", "--initial-velocity", "-tx", "-ty", "-push", "--push", "-omv", "--output-max-vel", "-ov",
                 "--output-velocity", "-ovf", "--output-velocity-field", "-f", "--file", "-sw",
                 "--shear-wave-decay", "-ld", "--lid-driven-cavity"):
        assert flag in h, flag


def test_parse_errors_print_message_and_usage_and_exit_1():
    p = run_main("-omv", "-ov", check=False)
    assert p.returncode == 1
    assert p.stderr.startswith("Argument '-ov/--output-velocity' not allowed with '-omv/--output-max-vel'\n")
    assert "Usage: lbm" in p.stderr
    p = run_main("-nx", "abc", check=False)
    assert p.returncode == 1 and p.stderr.startswith("pattern 'abc' not found\n")
    p = run_main("-sw", "-ld", check=False)
    assert p.returncode == 1 and "not allowed with '-sw/--shear-wave-decay'" in p.stderr
    p = run_main("--bogus", check=False)
    assert p.returncode == 1 and p.stderr.startswith("Unknown argument: --bogus\n")
    p = run_main("-nx", check=False)
    assert p.returncode == 1 and p.stderr.startswith("Too few arguments for '-nx'.")
    p = run_main("-f", "/nonexistent-dir/out.csv", "-s", "1", check=False)
    assert p.returncode == 1 and p.stderr.startswith("Failed to open file: /nonexistent-dir/out.csv")


# ---- on the GPU ---------------------------------------------------------------------------------
LAST_LINE = re.compile(r"^(\d+) lattice updates in \|(\d+)\|ns => (\d+) MLUPS$")


def check_stderr_contract(stderr, nx, ny, steps):
    lines = stderr.split("\n")
    assert lines[-1] == "" and lines[0] == f"running on{nx}x{ny}"
    m = LAST_LINE.match(lines[-2])
    assert m, lines[-3:]
    assert lines[-3] == ""                      # the blank line the reference prints before it
    assert int(m.group(1)) == nx * ny * steps
    ns = int(stderr.split("|")[-2])             # how benchmark.py:37 extracts it
    assert ns == int(m.group(2)) and ns > 0
    assert int(m.group(3)) == (nx * ny * steps * 1000) // ns


@pytest.mark.gpu
def test_omv_series_and_stderr_contract():
    p = run_main("-nx", 64, "-ny", 64, "-w", 0.5, "-s", 2001, "-of", 500, "-omv")
    _, s = orc.run("shear", "pull", 64, 64, 2001, 0.5, np.float32, out_every=500)
    assert p.stdout == "".join(orc.fmt(v) + "," for v in s)
    check_stderr_contract(p.stderr, 64, 64, 2001)
    # the long names address the tile sizes, only -nx/-ny the lattice (reference io.cpp:52,58)
    p = run_main("--x-grid-points", 8, "-ny", 8, "-s", 1)
    assert p.stderr.startswith("running on1024x8\n")


@pytest.mark.gpu
def test_push_flag_and_benchmark_command_shape():
    p = run_main("-nx", 256, "-ny", 128, "-of", 0, "-s", 100, "--shear-wave-decay", "-push")
    assert p.stdout == ""
    check_stderr_contract(p.stderr, 256, 128, 100)
    p = run_main("-nx", 96, "-ny", 40, "-w", 1.0, "-s", 301, "-of", 100, "-omv", "-push")
    _, s = orc.run("shear", "push", 96, 40, 301, 1.0, np.float32, out_every=100)
    assert p.stdout == "".join(orc.fmt(v) + "," for v in s)


@pytest.mark.gpu
def test_ov_and_ovf_blocks(tmp_path):
    p = run_main("-nx", 4, "-ny", 6, "-s", 1, "-of", 1, "-ov")
    f = orc.alloc(4, 6, np.float32)
    orc.init_shearwave(f, np.float32(1), np.float32(0.1))
    want = "\n".join(",".join(orc.fmt(v) for v in r) for r in orc.velocity_magnitudes(f)) + "\n#\n"
    assert p.stdout == want
    assert p.stdout.splitlines()[1] == "0.0866025,0.0866025,0.0866025,0.0866025"
    # lid-driven cavity, velocity field into a file, as lid-driven-stationary.py:41-54 calls it
    out = tmp_path / "vel.csv"
    p = run_main("-w", 1.7, "-nx", 24, "-ny", 16, "-of", 20, "-s", 41, "-ovf", "-ld", "-u", 0.1, "-rho", 1.0, "-f", out)
    assert p.stdout == ""
    want = ""
    for t in (0, 20, 40):
        fr, _ = orc.run("lid", "push", 24, 16, t, 1.7, np.float32)
        vel = orc.compute_velocities(fr)
        for c in range(2):
            want += "\n".join(",".join(orc.fmt(v) for v in r) for r in vel[c]) + "\n#\n"
    assert out.read_text() == want
    blocks = out.read_text().split("#")          # how the upstream scripts split frames
    assert len(blocks) == 7


@pytest.mark.gpu
def test_coalescing_false_build_prints_the_same(tmp_path):
    """build/main_aos is the driver compiled with COALESCING=false (reference macros.h:7): arrays are
    [NY][NX][9] throughout, the text that comes out is the same."""
    for args in (("-nx", 96, "-ny", 40, "-w", 1.0, "-s", 301, "-of", 100, "-omv", "-push"),
                 ("-nx", 130, "-ny", 70, "-w", 0.5, "-s", 1000, "-of", 250, "-omv"),
                 ("-nx", 24, "-ny", 16, "-w", 1.7, "-s", 41, "-of", 20, "-ovf", "-ld"),
                 ("-nx", 33, "-ny", 21, "-w", 1.2, "-s", 50, "-of", 25, "-ov", "-ld", "-push")):
        a = run_main(*args)
        b = run_main(*args, exe="main_aos")
        assert a.stdout == b.stdout and a.stdout != "", args
    p = run_main("-nx", 256, "-ny", 128, "-of", 0, "-s", 100, exe="main_aos")
    check_stderr_contract(p.stderr, 256, 128, 100)


@pytest.mark.gpu
def test_unroll_loops_false_build_follows_the_rolled_spelling():
    """build/main_rolled is the driver compiled with UNROLL_LOOPS=false (reference macros.h:6): the
    periodic sweeps use pull_periodic_for / push_periodic_for, the cavity is unchanged."""
    orc.set_rolled(True)
    try:
        for scheme, flag in (("pull", ()), ("push", ("-push",))):
            p = run_main("-nx", 96, "-ny", 40, "-w", 1.0, "-s", 301, "-of", 100, "-omv", *flag, exe="main_rolled")
            _, s = orc.run("shear", scheme, 96, 40, 301, 1.0, np.float32, out_every=100)
            assert p.stdout == "".join(orc.fmt(v) + "," for v in s)
        p = run_main("-nx", 64, "-ny", 48, "-w", 0.7, "-s", 120, "-of", 119, "-ov", exe="main_rolled")
        fr, _ = orc.run("shear", "pull", 64, 48, 119, 0.7, np.float32)
        start = orc.alloc(64, 48, np.float32)
        orc.init_shearwave(start, np.float32(1), np.float32(0.1))
        want = ""
        for state in (start, fr):
            want += "\n".join(",".join(orc.fmt(v) for v in r) for r in orc.velocity_magnitudes(state)) + "\n#\n"
        assert p.stdout == want
    finally:
        orc.set_rolled(False)
    args = ("-nx", 24, "-ny", 16, "-w", 1.7, "-s", 41, "-of", 20, "-ovf", "-ld")
    assert run_main(*args).stdout == run_main(*args, exe="main_rolled").stdout


@pytest.mark.gpu
def test_double_precision_build(golden_dir):
    """build/main_f64 is the driver compiled with USE_SINGLE_PRECISION=false (reference macros.h:5,
    global.h:21-27): the whole config-1 series in fp64 as the survey of the reference recorded it from
    the reference's sources compiled against a stand-in Kokkos header (a consistency check, not a
    pin: the reference's own build was never run)."""
    import json
    with open(os.path.join(golden_dir, "config1_series.json")) as fh:
        gold = json.load(fh)
    p = run_main("-nx", 1024, "-ny", 1024, "-w", 0.5, "-s", 50000, "-of", 500, "-omv", exe="main_f64")
    assert p.stdout == gold["fp64_text"].rstrip(",") + ","
    check_stderr_contract(p.stderr, 1024, 1024, 50000)
    p = run_main("-ld", "-nx", 24, "-ny", 16, "-w", 1.7, "-s", 41, "-of", 20, "-ovf", exe="main_f64")
    want = ""
    for t in (0, 20, 40):
        fr, _ = orc.run("lid", "push", 24, 16, t, 1.7, np.float64)
        vel = orc.compute_velocities(fr)
        for c in range(2):
            want += "\n".join(",".join(orc.fmt(v) for v in r) for r in vel[c]) + "\n#\n"
    assert p.stdout == want


@pytest.mark.gpu
def test_y_slabs_print_what_one_device_prints():
    """-gpus N (or LBM_GPUS=N): the lattice cut into N y-slabs driven by the one process, as the
    reference runs under `mpirun -n N` (main.cpp:92-401).  More slabs than devices share the devices,
    so one GPU exercises the whole schedule: same text as the single-slab run, for the periodic and
    the cavity scene, pull and push, slab heights that differ by a row, all three outputs."""
    cases = [("-nx", 256, "-ny", 203, "-w", 0.8, "-s", 1001, "-of", 100, "-omv"),
             ("-nx", 256, "-ny", 203, "-w", 0.8, "-s", 1001, "-of", 100, "-omv", "-push"),
             ("-nx", 96, "-ny", 70, "-w", 1.2, "-s", 201, "-of", 100, "-ovf", "-ld"),
             ("-nx", 64, "-ny", 61, "-w", 1.0, "-s", 50, "-of", 7, "-ov"),
             ("-nx", 1024, "-ny", 2048, "-w", 0.5, "-s", 300, "-of", 50, "-omv")]
    for args in cases:
        one = run_main(*args)
        for n in (2, 3, 7):
            p = run_main(*args, "-gpus", n)
            assert p.stdout == one.stdout, (args, n)
            lines = p.stderr.split("\n")
            assert lines[1].startswith(f"{n} y-slab(s) of "), lines[:3]
            nx, ny, steps = args[1], args[3], args[7]
            check_stderr_contract(p.stderr, nx, ny, steps)
    # the environment variable does the same; slabs thinner than two rows are refused
    env = dict(os.environ, LBM_GPUS="4")
    p = subprocess.run(["./main", *map(str, cases[0])], cwd=BUILD, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and p.stdout == run_main(*cases[0]).stdout and "4 y-slab(s)" in p.stderr
    p = run_main("-nx", 64, "-ny", 9, "-s", 4, "-gpus", 8, check=False)
    assert p.returncode == 1 and "too thin" in p.stderr
    # the contracted arithmetic: slabs against one device, and
    # one device against the oracle's restatement of the contraction
    env = dict(os.environ, LBM_ARITHMETIC="contracted")
    args = ("-nx", 512, "-ny", 300, "-w", 1.1, "-s", 401, "-of", 100, "-omv")
    runs = [subprocess.run(["./main", *map(str, args), *extra], cwd=BUILD, capture_output=True, text=True, timeout=600, env=env)
            for extra in ((), ("-gpus", "3"), ("-gpus", "5"))]
    assert all(r.returncode == 0 for r in runs), [r.stderr for r in runs]
    assert runs[1].stdout == runs[0].stdout and runs[2].stdout == runs[0].stdout
    assert "3 y-slab(s) of 100 rows" in runs[1].stderr
    with orc.arithmetic("contracted"):
        _, s = orc.run("shear", "pull", 512, 300, 401, 1.1, np.float32, out_every=100)
    assert runs[0].stdout == "".join(orc.fmt(v) + "," for v in s)


@pytest.mark.gpu
def test_y_slabs_over_rccl_print_what_one_device_prints():
    """LBM_SLAB_TRANSPORT=rccl: the exchange as ncclSend / ncclRecv pairs inside one group on the
    devices' edge streams (one communicator per device, made by the one process).  On one GPU the
    device is its own peer and every slab's messages go through the communicator to itself, which
    exercises staging, the pairing of several messages between the same two ranks, and the
    schedule; the text must be that of the single-slab run, and of the copy transport."""
    cases = [("-nx", 256, "-ny", 203, "-w", 0.8, "-s", 1001, "-of", 100, "-omv"),
             ("-nx", 256, "-ny", 203, "-w", 0.8, "-s", 401, "-of", 100, "-omv", "-push"),
             ("-nx", 96, "-ny", 70, "-w", 1.2, "-s", 201, "-of", 100, "-ovf", "-ld"),
             ("-nx", 64, "-ny", 61, "-w", 1.0, "-s", 50, "-of", 7, "-ov"),
             ("-nx", 1024, "-ny", 2048, "-w", 0.5, "-s", 300, "-of", 50, "-omv")]
    env = dict(os.environ, LBM_SLAB_TRANSPORT="rccl")
    for args in cases:
        one = run_main(*args)
        for n in (2, 3, 7):
            p = subprocess.run(["./main", *map(str, args), "-gpus", str(n)], cwd=BUILD, capture_output=True, text=True,
                               timeout=600, env=env)
            assert p.returncode == 0, p.stderr
            assert p.stdout == one.stdout, (args, n)
            assert "RCCL send/recv" in p.stderr.split("\n")[1], p.stderr
            check_stderr_contract(p.stderr, args[1], args[3], args[7])
    # the default on one device is the copy transport; an unknown name is refused
    p = run_main(*cases[0], "-gpus", 2)
    assert "device-to-device copies" in p.stderr.split("\n")[1]
    bad = subprocess.run(["./main", *map(str, cases[0]), "-gpus", "2"], cwd=BUILD, capture_output=True, text=True,
                         timeout=600, env=dict(os.environ, LBM_SLAB_TRANSPORT="mpi"))
    assert bad.returncode == 1 and "LBM_SLAB_TRANSPORT" in bad.stderr


@pytest.mark.gpu
def test_two_d_grid_of_blocks_prints_what_one_device_prints():
    """LBM_GRID=PXxPY / LBM_DECOMPOSITION=2d: the periodic lattice cut in both directions as the
    reference cuts it under MPI (main.cpp:96-144), blocks on frames of up to eight nodes advanced several
    steps per launch.  All blocks share the one GPU here; the text must be that of the undivided run:
    pull and push, block extents that differ by a node, rows of odd and even length, all outputs,
    both arithmetics."""
    cases = [("-nx", 256, "-ny", 203, "-w", 0.8, "-s", 401, "-of", 100, "-omv"),
             ("-nx", 250, "-ny", 120, "-w", 0.8, "-s", 301, "-of", 100, "-omv", "-push"),
             ("-nx", 67, "-ny", 61, "-w", 1.0, "-s", 50, "-of", 7, "-ov"),
             ("-nx", 96, "-ny", 70, "-w", 1.2, "-s", 120, "-of", 40, "-ovf", "-push"),
             ("-nx", 1024, "-ny", 1024, "-w", 0.5, "-s", 200, "-of", 50, "-omv")]
    for arith in ("strict", "contracted"):
        base_env = dict(os.environ, LBM_ARITHMETIC=arith)
        for args in cases:
            one = subprocess.run(["./main", *map(str, args)], cwd=BUILD, capture_output=True, text=True, timeout=600, env=base_env)
            assert one.returncode == 0, one.stderr
            for grid in ("2x1", "1x3", "2x2", "3x2", "4x2"):
                p = subprocess.run(["./main", *map(str, args)], cwd=BUILD, capture_output=True, text=True, timeout=600,
                                   env=dict(base_env, LBM_GRID=grid))
                assert p.returncode == 0, (grid, p.stderr)
                assert p.stdout == one.stdout, (arith, args, grid)
                assert p.stderr.split("\n")[1].startswith(f"{grid} block(s) of "), p.stderr
                check_stderr_contract(p.stderr, args[1], args[3], args[7])
    # -gpus N with LBM_DECOMPOSITION=2d takes the grid MPI_Dims_create would: 6 -> 3 x 2
    env = dict(os.environ, LBM_DECOMPOSITION="2d")
    p = subprocess.run(["./main", *map(str, cases[0]), "-gpus", "6"], cwd=BUILD, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "3x2 block(s) of " in p.stderr and p.stdout == run_main(*cases[0]).stdout
    # the cavity is refused, as under the reference's MPI build (main.cpp:317-319)
    p = subprocess.run(["./main", "-nx", "64", "-ny", "64", "-s", "10", "-ld"], cwd=BUILD, capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, LBM_GRID="2x2"))
    assert p.returncode == 1 and "periodic scene only" in p.stderr
    p = subprocess.run(["./main", "-nx", "64", "-ny", "64", "-s", "10"], cwd=BUILD, capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, LBM_GRID="two by two"))
    assert p.returncode == 1 and "LBM_GRID" in p.stderr


@pytest.mark.gpu
def test_ov_output_stops_at_the_first_nan():
    """io.cpp:239-243: the -ov output prints values in scan order until it meets a NaN, reports its
    place on stderr and the run ends with exit code 1.  An unstable cavity (lid at Mach 1.7) provides
    the NaN; text and place must be those of the oracle's run (strict arithmetic: the suite's default)."""
    nx, ny, omega, u = 32, 24, 1.95, 1.0
    p = run_main("-nx", nx, "-ny", ny, "-w", omega, "-u", u, "-s", 200, "-of", 10, "-ov", "-ld", check=False)
    f, buf = orc.alloc(nx, ny, np.float32), orc.alloc(nx, ny, np.float32)
    orc.init_rest(f, np.float32(1.0))
    want, place = "", None
    for t in range(200):
        if t % 10 == 0:
            m = orc.velocity_magnitudes(f)
            if np.isnan(m).any():
                ys, xs = np.nonzero(np.isnan(m))
                y, x = int(ys[0]), int(xs[0])
                for row in m[:y]:
                    want += ",".join(orc.fmt(v) for v in row) + "\n"
                want += "".join(orc.fmt(v) + "," for v in m[y, :x])
                place = (x, y)
                break
            want += "\n".join(",".join(orc.fmt(v) for v in row) for row in m) + "\n#\n"
        orc.push_lid_driven(f, buf, np.float32(omega), np.float32(1.0), np.float32(u))
        f, buf = buf, f
    assert place is not None, "the oracle's run stayed finite: pick other parameters"
    assert p.returncode == 1
    assert f"NaN encountered at x={place[0]} y={place[1]}" in p.stderr
    assert p.stdout == want
