"""HIP path (through the C ABI) against the CPU oracle, bit for bit.

Every comparison is exact (integer view of the IEEE bits): the kernels are compiled without FMA
contraction and follow the reference's expression trees, so they must reproduce the strict oracle
to the last bit -- far inside the path's stated tolerance of 1e-12 (fp64) / 1e-5 (fp32) max-abs on
populations.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

TORCH = {np.float32: torch.float32, np.float64: torch.float64}
BITS = {np.float32: np.uint32, np.float64: np.uint64}


def assert_bits_equal(got, want, what=""):
    got = np.ascontiguousarray(got)
    want = np.ascontiguousarray(want)
    assert got.dtype == want.dtype and got.shape == want.shape, (got.dtype, want.dtype, got.shape, want.shape)
    u = BITS[got.dtype.type]
    # NaN payload/sign bits are not specified by IEEE 754 propagation rules: NaN == NaN here
    bad = (got.view(u) != want.view(u)) & ~(np.isnan(got) & np.isnan(want))
    if bad.any():
        idx = np.argwhere(bad)[0]
        diff = np.nanmax(np.abs(got.astype(np.float64) - want.astype(np.float64)))
        raise AssertionError(f"{what}: {bad.sum()} of {bad.size} values differ, first at {tuple(idx)}: "
                             f"{got[tuple(idx)]!r} vs {want[tuple(idx)]!r}; max abs diff {diff:g}")


def gpu_sim(scene, scheme, nx, ny, dtype, omega):
    from lbm_amd import Simulation
    return Simulation(nx, ny, scene=scene, scheme=scheme, dtype=TORCH[dtype], omega=omega, device="cuda:0")


# (nx, ny): multiples of the vector width, odd / non-multiple row lengths (element-aligned vector
# kernel), rows shorter than four vectors (one-node-per-lane kernel), degenerate lattices
SIZES = [(8, 6), (64, 48), (100, 60), (128, 96), (7, 5), (1030, 9), (2048, 4), (4, 3), (1, 1), (2, 7),
         (67, 12), (1027, 5), (262, 9), (17, 4), (513, 3)]
CASES = [("shear", "pull"), ("shear", "push"), ("lid", "push"), ("lid", "pull")]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scene,scheme", CASES)
@pytest.mark.parametrize("nx,ny", SIZES)
@pytest.mark.parametrize("max_fuse", [1, None], ids=["one-step", "fused"])
def test_populations_match_oracle_bit_for_bit(dtype, scene, scheme, nx, ny, max_fuse):
    """max_fuse=1: one launch of the one-step sweep per time step; None: the time loop's default
    (runs of up to six steps per fused launch where the row length allows it)."""
    for omega in (0.5, 1.0, 1.7):
        sim = gpu_sim(scene, scheme, nx, ny, dtype, omega)
        ref, _ = orc.run(scene, scheme, nx, ny, 0, omega, dtype)
        assert_bits_equal(sim.f.cpu().numpy(), ref, f"init {scene} {nx}x{ny}")
        done = 0
        for upto in (1, 2, 10, 60):
            sim.step(upto - done, max_fuse=max_fuse)
            done = upto
            ref, _ = orc.run(scene, scheme, nx, ny, upto, omega, dtype)
            assert_bits_equal(sim.f.cpu().numpy(), ref, f"{scene}/{scheme} {nx}x{ny} w={omega} t={upto}")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scene,scheme", CASES)
def test_long_run_1000_steps(dtype, scene, scheme, monkeypatch):
    nx, ny, omega = 128, 96, 0.5
    ref, _ = orc.run(scene, scheme, nx, ny, 1000, omega, dtype)
    sim = gpu_sim(scene, scheme, nx, ny, dtype, omega)
    sim.step(1000)
    assert sim.launches == 200   # runs of five steps on a lattice this small
    assert_bits_equal(sim.f.cpu().numpy(), ref, f"{scene}/{scheme} 1000 steps, fused launches of 5")
    # the same run with a size threshold that sends this lattice to the one-step sweeps
    from lbm_amd import ops
    with ops.tuning("fused_min_nodes", 1000000):
        sim = gpu_sim(scene, scheme, nx, ny, dtype, omega)
        sim.step(1000)
        assert sim.launches == 1000
        assert_bits_equal(sim.f.cpu().numpy(), ref, f"{scene}/{scheme} 1000 steps, one-step launches")
    with ops.tuning("fused_max_steps", 3):
        sim = gpu_sim(scene, scheme, nx, ny, dtype, omega)
        sim.step(1000)
        assert_bits_equal(sim.f.cpu().numpy(), ref, f"{scene}/{scheme} 1000 steps, fused launches of 3")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_random_state_single_sweep(dtype):
    """Arbitrary (non-equilibrium, non-symmetric) populations: catches any swapped channel or
    neighbour, which the x-invariant shear wave cannot."""
    from lbm_amd import ops
    rng = np.random.default_rng(1234)
    for nx, ny in ((64, 33), (256, 17), (1024, 5), (37, 11), (4096, 3), (1029, 7), (130, 9), (22627, 3)):
        f = (1.0 / 9 + 0.02 * rng.standard_normal((9, ny, nx))).astype(dtype)
        d_f = torch.from_numpy(f).cuda()
        d_b = torch.full_like(d_f, float("nan"))
        om, rho, u = dtype(1.3), dtype(1.0), dtype(0.1)
        for name in ("pull_periodic_no_mpi", "push_periodic_no_mpi"):
            want = np.full_like(f, np.nan)
            getattr(orc, name)(f, want, om)
            d_b.fill_(float("nan"))
            getattr(ops, name)(d_f, d_b, nx, ny, float(om))
            assert_bits_equal(d_b.cpu().numpy(), want, f"{name} {nx}x{ny}")
        for name in ("push_lid_driven", "pull_lid_driven"):
            want = np.full_like(f, np.nan)
            getattr(orc, name)(f, want, om, rho, u)
            d_b.fill_(float("nan"))
            getattr(ops, name)(d_f, d_b, nx, ny, float(om), 1.0, 0.1)
            assert_bits_equal(d_b.cpu().numpy(), want, f"{name} {nx}x{ny}")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_halo_forms_match_oracle(dtype):
    """pull/push_periodic_inner + _outer on a halo'd array, and the y-slab rows form."""
    from lbm_amd import ops
    rng = np.random.default_rng(99)
    nx, ny = 70, 23
    f = (1.0 / 9 + 0.02 * rng.standard_normal((9, ny, nx))).astype(dtype)
    om = dtype(0.9)
    d_f = torch.from_numpy(f).cuda()
    for kind in ("pull", "push"):
        want = np.zeros_like(f)
        getattr(orc, f"{kind}_periodic_inner")(f, want, om, 2, 2, nx - 2, ny - 2)
        getattr(orc, f"{kind}_periodic_outer")(f, want, om)
        d_b = torch.zeros_like(d_f)
        getattr(ops, f"{kind}_periodic_inner")(d_f, d_b, float(om), 2, 2, nx - 2, ny - 2)
        getattr(ops, f"{kind}_periodic_outer")(d_f, d_b, float(om), nx, ny)
        assert_bits_equal(d_b.cpu().numpy(), want, f"{kind} inner+outer")
    # rectangles of every width around the change from the element-wise to the vector kernel, odd
    # bounds, single rows; nothing outside the rectangle may be written
    for nx, ny, LX, LY, HX, HY in ((70, 23, 1, 1, 69, 22), (70, 23, 3, 5, 64, 6), (40, 9, 2, 2, 17, 7),
                                   (40, 9, 2, 2, 18, 7), (40, 9, 5, 1, 24, 8), (1031, 12, 1, 1, 1030, 11),
                                   (1031, 12, 7, 3, 1000, 9), (12, 40, 1, 1, 11, 39)):
        f = (1.0 / 9 + 0.02 * rng.standard_normal((9, ny, nx))).astype(dtype)
        d_f = torch.from_numpy(f).cuda()
        for kind in ("pull", "push"):
            want = np.full_like(f, -7.0)
            getattr(orc, f"{kind}_periodic_inner")(f, want, om, LX, LY, HX, HY)
            d_b = torch.full_like(d_f, -7.0)
            getattr(ops, f"{kind}_periodic_inner")(d_f, d_b, float(om), LX, LY, HX, HY)
            assert_bits_equal(d_b.cpu().numpy(), want, f"{kind} inner {nx}x{ny} [{LX},{HX})x[{LY},{HY})")
    nx, ny = 256, 23
    f = (1.0 / 9 + 0.02 * rng.standard_normal((9, ny, nx))).astype(dtype)
    d_f = torch.from_numpy(f).cuda()
    for kind in ("pull", "push"):
        want = np.zeros_like(f)
        getattr(orc, f"{kind}_rows")(f, want, om, 1, ny - 1, False)
        d_b = torch.zeros_like(d_f)
        getattr(ops, f"{kind}_periodic_rows")(d_f, d_b, float(om), 1, ny - 1, False)
        assert_bits_equal(d_b.cpu().numpy(), want, f"{kind} rows (ghost rows)")
        want = np.zeros_like(f)
        getattr(orc, f"{kind}_rows")(f, want, om, 3, 9, True)
        d_b = torch.zeros_like(d_f)
        getattr(ops, f"{kind}_periodic_rows")(d_f, d_b, float(om), 3, 9, True)
        assert_bits_equal(d_b.cpu().numpy(), want, f"{kind} rows (wrap)")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_output_path_matches_oracle(dtype):
    from lbm_amd import ops
    rng = np.random.default_rng(5)
    for nx, ny in ((64, 48), (101, 7), (2048, 3)):
        f = (1.0 / 9 + 0.02 * rng.standard_normal((9, ny, nx))).astype(dtype)
        f[3, ny // 2, nx // 3] = np.nan  # NaNs are skipped by the max (io.cpp:205)
        d_f = torch.from_numpy(f).cuda()
        vel = torch.empty((2, ny, nx), dtype=TORCH[dtype], device="cuda")
        ops.compute_velocities(vel, d_f, 0, 0, nx, ny)
        assert_bits_equal(vel.cpu().numpy(), orc.compute_velocities(f), "velocities")
        mag = torch.empty((ny, nx), dtype=TORCH[dtype], device="cuda")
        ops.velocity_magnitudes(mag, d_f, 0, ny)
        assert_bits_equal(mag.cpu().numpy(), orc.velocity_magnitudes(f), "magnitudes")
        got = ops.max_velocity(d_f)
        assert got == float(orc.max_velocity(f)), (got, orc.max_velocity(f))
        if ny > 4:
            got = ops.max_velocity(d_f, 1, ny - 2)
            assert got == float(orc.max_velocity(f, 1, ny - 2))
    f[:] = np.nan
    assert ops.max_velocity(torch.from_numpy(f).cuda()) == float(dtype(-1e30))


def test_text_outputs_match_oracle():
    """-omv / -ov / -ovf text as the reference prints it (io.cpp:219, :234-254, :268-286)."""
    import io
    from lbm_amd import Simulation
    # -nx 64 -ny 64 -w 0.5 -s 2001 -of 500 -omv  (fp64: recorded in the reference survey probe)
    sim = Simulation(64, 64, dtype=torch.float64, omega=0.5)
    assert sim.run(2001, 500, "omv") == "0.1,0.00918508,0.000838386,7.65257e-05,6.98548e-06,"
    sim = Simulation(4, 6, dtype=torch.float32)
    text = sim.run(1, 1, "ov")
    lines = text.splitlines()
    assert len(lines) == 7 and lines[-1] == "#"
    assert lines[1] == "0.0866025,0.0866025,0.0866025,0.0866025"
    f = orc.alloc(4, 6, np.float32)
    orc.init_shearwave(f, np.float32(1), np.float32(0.1))
    want = "\n".join(",".join(orc.fmt(v) for v in r) for r in orc.velocity_magnitudes(f)) + "\n#\n"
    assert text == want
    sim = Simulation(8, 6, scene="lid", scheme="push", dtype=torch.float32, omega=1.7)
    sim.step(25)
    buf = io.StringIO()
    sim.output("ovf", buf)
    fr, _ = orc.run("lid", "push", 8, 6, 25, 1.7, np.float32)
    vel = orc.compute_velocities(fr)
    want = ""
    for c in range(2):
        want += "\n".join(",".join(orc.fmt(v) for v in r) for r in vel[c]) + "\n#\n"
    assert buf.getvalue() == want


def test_slab_simulation_single_rank_equals_plain_simulation():
    from lbm_amd import Simulation, SlabSimulation
    for scene, scheme in CASES:
        for dtype in (torch.float32, torch.float64):
            a = Simulation(256, 40, scene=scene, scheme=scheme, dtype=dtype, omega=1.1)
            a.step(37)
            for halo in (1, 2, 4, 6):   # one halo row per step; deep halos, fused blocks of that many steps
                b = SlabSimulation(256, 40, scene=scene, scheme=scheme, dtype=dtype, omega=1.1, halo=halo)
                b.step(37)
                assert torch.equal(a.f, b.interior()), (scene, scheme, dtype, halo)
                assert a.max_velocity() == b.max_velocity()


def test_invalid_arguments_are_rejected():
    from lbm_amd import capi, ops
    f = torch.zeros((9, 4, 8), dtype=torch.float32, device="cuda")
    with pytest.raises(capi.LbmError):
        ops.pull_periodic_no_mpi(f, f, 8, 4, 0.5)            # f and buf overlap
    with pytest.raises(capi.LbmError):
        ops.pull_periodic_rows(f, torch.zeros_like(f), 0.5, 0, 4, False)   # no ghost rows
    with pytest.raises(capi.LbmError):
        ops.pull_periodic_inner(f, torch.zeros_like(f), 0.5, 0, 1, 8, 3)   # no x halo
    with pytest.raises(RuntimeError):
        ops.init_rest(torch.zeros((9, 4, 8)), 1.0)           # CPU tensor: no fallback
    with pytest.raises(ValueError):
        ops.init_rest(torch.zeros((8, 4, 8), device="cuda"), 1.0)
