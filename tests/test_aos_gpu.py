"""The reference's other memory layout, [NY][NX][9] (COALESCING=false, global.h:40, main.cpp:34-35).

Every operator must give, node by node, the bits it gives on [9][NY][NX] arrays; those are pinned
against the CPU oracle elsewhere, and the sweeps are compared with the oracle directly here too.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from test_parity_gpu import assert_bits_equal
from test_fused_gpu import ONE_STEP, RHO, U, SIZES, oracle_steps, random_state

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def aos_layout():
    from lbm_amd import ops
    with ops.layout("aos"):
        yield
    assert ops.get_layout() == "soa"


def to_aos(a):
    """[9, NY, NX] -> contiguous [NY, NX, 9]"""
    return np.ascontiguousarray(np.transpose(a, (1, 2, 0)))


def to_soa(a):
    return np.ascontiguousarray(np.transpose(a, (2, 0, 1)))


def run_fused(scene, scheme, d_f, d_b, om, y_lo, y_hi, k, wrap_y=True, y_goff=0, ny_glob=None):
    from lbm_amd import ops
    if scene == "shear":
        getattr(ops, f"{scheme}_periodic_fused")(d_f, d_b, om, y_lo, y_hi, wrap_y, k)
    else:
        getattr(ops, f"{scheme}_lid_fused")(d_f, d_b, om, RHO, U, y_lo, y_hi, y_goff, ny_glob, k)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scene,scheme", list(ONE_STEP))
@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6])
def test_whole_lattice_matches_k_oracle_steps(dtype, scene, scheme, k):
    rng = np.random.default_rng(17 * k + (dtype == np.float64))
    for nx, ny in SIZES:
        if scene == "lid" and (ny < 2 or nx < 2):
            continue
        f = random_state(rng, nx, ny, dtype)
        om = 1.3
        want = oracle_steps(ONE_STEP[(scene, scheme)], f, k, om, dtype)
        d_f = torch.from_numpy(to_aos(f)).cuda()
        d_b = torch.full_like(d_f, float("nan"))
        run_fused(scene, scheme, d_f, d_b, om, 0, ny, k, wrap_y=True, y_goff=0, ny_glob=ny)
        assert_bits_equal(to_soa(d_b.cpu().numpy()), want, f"aos {scene}/{scheme} k={k} {nx}x{ny}")
        assert torch.equal(d_f.cpu(), torch.from_numpy(to_aos(f))), "the input array was modified"


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scene,scheme", list(ONE_STEP))
def test_one_step_entry_points(dtype, scene, scheme):
    """The reference-named whole-lattice sweeps (pull-push.h:73-140) on [NY][NX][9] arrays."""
    from lbm_amd import ops
    rng = np.random.default_rng(5)
    for nx, ny in ((64, 48), (229, 11), (1024, 70)):
        f = random_state(rng, nx, ny, dtype)
        want = oracle_steps(ONE_STEP[(scene, scheme)], f, 1, 0.9, dtype)
        d_f = torch.from_numpy(to_aos(f)).cuda()
        d_b = torch.full_like(d_f, float("nan"))
        if scene == "shear":
            getattr(ops, ONE_STEP[(scene, scheme)])(d_f, d_b, nx, ny, 0.9)
        else:
            getattr(ops, ONE_STEP[(scene, scheme)])(d_f, d_b, nx, ny, 0.9, RHO, U)
        assert_bits_equal(to_soa(d_b.cpu().numpy()), want, f"aos {scene}/{scheme} one step {nx}x{ny}")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scheme", ["pull", "push"])
@pytest.mark.parametrize("k", [1, 3, 6])
def test_row_ranges_with_ghost_rows(dtype, scheme, k):
    rng = np.random.default_rng(100 + k)
    nx, ny = 256, 40
    f = random_state(rng, nx, ny, dtype)
    want = oracle_steps(f"{scheme}_periodic_no_mpi", f, k, 0.8, dtype)
    d_f = torch.from_numpy(to_aos(f)).cuda()
    d_b = torch.full_like(d_f, -7.0)
    run_fused("shear", scheme, d_f, d_b, 0.8, k, ny - k, k, wrap_y=False)
    got = to_soa(d_b.cpu().numpy())
    assert_bits_equal(got[:, k:ny - k], want[:, k:ny - k], f"aos {scheme} rows k={k}")
    assert (got[:, :k] == -7.0).all() and (got[:, ny - k:] == -7.0).all()
    # a slab of a taller cavity, under the lid
    want = oracle_steps(f"{scheme}_lid_driven", f, k, 1.6, dtype)
    g0, g1 = 27, 40
    lo = g0 - k
    local = np.ascontiguousarray(f[:, lo:])
    d_f = torch.from_numpy(to_aos(local)).cuda()
    d_b = torch.full_like(d_f, -7.0)
    run_fused("lid", scheme, d_f, d_b, 1.6, g0 - lo, g1 - lo, k, y_goff=lo, ny_glob=ny)
    got = to_soa(d_b.cpu().numpy())
    assert_bits_equal(got[:, g0 - lo:], want[:, g0:g1], f"aos lid {scheme} slab k={k}")
    assert (got[:, :g0 - lo] == -7.0).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_initialisers_and_outputs_equal_the_soa_ones(dtype):
    from lbm_amd import ops
    nx, ny = 200, 37
    with ops.layout("soa"):
        s = torch.full(ops.population_shape(nx, ny), float("nan"), dtype=dtype, device="cuda")
        ops.init_shearwave(s, 1.0, 0.1)
        rest = torch.empty_like(s)
        ops.init_rest(rest, 1.25)
        vel_s = torch.empty((2, ny, nx), dtype=dtype, device="cuda")
        ops.compute_velocities(vel_s, s, 0, 0, nx, ny)
        mag_s = torch.empty((ny - 3, nx), dtype=dtype, device="cuda")
        ops.velocity_magnitudes(mag_s, s, 2, ny - 1)
        max_s = ops.max_velocity(s)
    a = torch.full(ops.population_shape(nx, ny), float("nan"), dtype=dtype, device="cuda")
    assert tuple(a.shape) == (ny, nx, 9)
    ops.init_shearwave(a, 1.0, 0.1)
    assert torch.equal(a.permute(2, 0, 1), s)
    rest_a = torch.empty_like(a)
    ops.init_rest(rest_a, 1.25)
    assert torch.equal(rest_a.permute(2, 0, 1), rest)
    vel_a = torch.empty((2, ny, nx), dtype=dtype, device="cuda")
    ops.compute_velocities(vel_a, a, 0, 0, nx, ny)
    assert torch.equal(vel_a, vel_s)
    mag_a = torch.empty((ny - 3, nx), dtype=dtype, device="cuda")
    ops.velocity_magnitudes(mag_a, a, 2, ny - 1)
    assert torch.equal(mag_a, mag_s)
    assert ops.max_velocity(a) == max_s
    # halo'd initialisation leaves the ghost rows alone
    h = torch.full((ny, nx, 9), -3.0, dtype=dtype, device="cuda")
    ops.init_shearwave(h, 1.0, 0.1, y_below=5, y_glob=100, halo=2)
    assert bool((h[:2] == -3.0).all()) and bool((h[-2:] == -3.0).all()) and bool((h[2:-2] != -3.0).all())


def test_unsupported_forms_say_so():
    from lbm_amd import capi, ops
    f = torch.zeros((12, 64, 9), dtype=torch.float32, device="cuda")
    b = torch.zeros_like(f)
    with pytest.raises(capi.LbmError, match="unsupported"):
        ops.push_periodic_rows(f, b, 0.7, 1, 11, False)
    with pytest.raises(capi.LbmError, match="unsupported"):
        ops.push_lid_rows(f, b, 0.7, 1.0, 0.1, 0, 6, 0, 12)
    with pytest.raises(capi.LbmError, match="unsupported"):
        ops.pull_periodic_inner(f, b, 0.7, 1, 1, 63, 11)
    with pytest.raises(capi.LbmError, match="unsupported"):
        ops.push_periodic_outer(f, b, 0.7, 64, 12)
    with pytest.raises(ValueError):
        ops.pull_periodic_no_mpi(torch.zeros((9, 12, 64), device="cuda"), torch.zeros((9, 12, 64), device="cuda"), 64, 12, 0.7)
    # pull row ranges keep their meaning
    rng = np.random.default_rng(1)
    s = random_state(rng, 64, 12, np.float32)
    want = oracle_steps("pull_periodic_no_mpi", s, 1, 0.7, np.float32)
    d_f = torch.from_numpy(to_aos(s)).cuda()
    d_b = torch.full_like(d_f, -7.0)
    ops.pull_periodic_rows(d_f, d_b, 0.7, 1, 11, False)
    got = to_soa(d_b.cpu().numpy())
    assert_bits_equal(got[:, 1:11], want[:, 1:11], "aos pull rows")
    assert (got[:, 0] == -7.0).all() and (got[:, 11] == -7.0).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_large_lattice_six_steps_equal_the_soa_result(dtype):
    from lbm_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(5)
    nx, ny = 4096, 1500
    s = 1.0 / 9 + 0.02 * torch.randn((9, ny, nx), dtype=dtype, device="cuda", generator=gen)
    a = s.permute(1, 2, 0).contiguous()
    for scene in ("shear", "lid"):
        for scheme in ("pull", "push"):
            want = torch.empty_like(s)
            with ops.layout("soa"):
                run_fused(scene, scheme, s, want, 1.1, 0, ny, 6, ny_glob=ny)
            got = torch.full_like(a, float("nan"))
            run_fused(scene, scheme, a, got, 1.1, 0, ny, 6, ny_glob=ny)
            assert torch.equal(got.permute(2, 0, 1), want), (scene, scheme)


@pytest.mark.parametrize("scene,scheme", [("shear", "pull"), ("shear", "push"), ("lid", "push"), ("lid", "pull")])
def test_simulation_runs_the_same_in_both_layouts(scene, scheme):
    from lbm_amd import ops
    from lbm_amd.sim import Simulation
    a = Simulation(300, 170, scene=scene, scheme=scheme, omega=1.2)
    assert tuple(a.f.shape) == (170, 300, 9)
    a.step(137)
    vel_a, max_a = a.velocities().clone(), a.max_velocity()
    with ops.layout("soa"):
        s = Simulation(300, 170, scene=scene, scheme=scheme, omega=1.2)
        s.step(137)
        assert torch.equal(s.velocities(), vel_a) and s.max_velocity() == max_a
        assert torch.equal(a.f.permute(2, 0, 1), s.f)
    assert a.launches == s.launches


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scene,scheme", list(ONE_STEP))
@pytest.mark.parametrize("k", [1, 2, 5, 7, 8])
def test_contracted_arithmetic_matches_its_oracle_restatement(dtype, scene, scheme, k):
    """LBM_ARITH_CONTRACTED on [NY][NX][9] arrays: the bits of the oracle's operation-by-operation
    restatement of the explicit contraction, as on [9][NY][NX] arrays; fp32 rows of an odd length
    are refused."""
    from lbm_amd import capi, ops
    rng = np.random.default_rng(900 + k)
    with orc.arithmetic("contracted"), ops.arithmetic("contracted"):
        for nx, ny in SIZES:
            if scene == "lid" and (ny < 2 or nx < 2):
                continue
            f = random_state(rng, nx, ny, dtype)
            d_f = torch.from_numpy(to_aos(f)).cuda()
            d_b = torch.full_like(d_f, float("nan"))
            if dtype == np.float32 and (nx % 2 or nx < 4):
                with pytest.raises(capi.LbmError, match="unsupported"):
                    run_fused(scene, scheme, d_f, d_b, 1.3, 0, ny, k, wrap_y=True, y_goff=0, ny_glob=ny)
                continue
            want = oracle_steps(ONE_STEP[(scene, scheme)], f, k, 1.3, dtype)
            run_fused(scene, scheme, d_f, d_b, 1.3, 0, ny, k, wrap_y=True, y_goff=0, ny_glob=ny)
            assert_bits_equal(to_soa(d_b.cpu().numpy()), want, f"aos contracted {scene}/{scheme} k={k} {nx}x{ny}")
        # the reference-named whole-lattice sweep
        nx, ny = 230, 11
        f = random_state(rng, nx, ny, dtype)
        want = oracle_steps(ONE_STEP[(scene, scheme)], f, 1, 0.9, dtype)
        d_f = torch.from_numpy(to_aos(f)).cuda()
        d_b = torch.full_like(d_f, float("nan"))
        if scene == "shear":
            getattr(ops, ONE_STEP[(scene, scheme)])(d_f, d_b, nx, ny, 0.9)
        else:
            getattr(ops, ONE_STEP[(scene, scheme)])(d_f, d_b, nx, ny, 0.9, RHO, U)
        assert_bits_equal(to_soa(d_b.cpu().numpy()), want, f"aos contracted {scene}/{scheme} one step")
    assert ops.get_arithmetic() == "unrolled" and orc.get_arithmetic() == "unrolled"
