"""LBM_ARITH_CONTRACTED: the unrolled relaxation with explicit fused multiply-adds where the
reference's own build flags let a compiler fuse (CMakeLists.txt:18; pull-push.cpp:25-72).

* bit for bit against the oracle's operation-by-operation restatement (collide_fma), every sweep,
  one step and up to eight steps per call;
* within north_star's parity budget of the strict spelling after thousands of steps (1e-5 fp32,
  1e-12 fp64, max-abs on populations);
* against the decay series the reference's authors recorded (docs/decays.js), whose build contracted;
* the fp32 reciprocal (hardware estimate + one fused Newton step) equals the correctly rounded
  quotient for every fp32 value of three binades, the fp64 one for millions of sampled values.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from test_parity_gpu import assert_bits_equal
from test_fused_gpu import ONE_STEP, SIZES, oracle_steps, random_state, run_fused, RHO, U

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def contracted_arithmetic():
    from lbm_amd import ops
    with orc.arithmetic("contracted"):
        with ops.arithmetic("contracted"):
            yield
    assert ops.get_arithmetic() == "unrolled" and orc.get_arithmetic() == "unrolled"


def test_reciprocal_is_correctly_rounded_for_every_fp32_of_three_binades():
    """1/x scales exactly with the exponent of x, so [0.5, 4) speaks for every normal x whose
    reciprocal is normal."""
    from lbm_amd import ops
    lo, hi = np.float32(0.5).view(np.uint32), np.float32(4.0).view(np.uint32)
    bad = 0
    for start in range(int(lo), int(hi), 1 << 23):
        bits = np.arange(start, min(start + (1 << 23), int(hi)), dtype=np.uint32)
        x = bits.view(np.float32)
        got = ops.selftest_reciprocal(torch.from_numpy(x).cuda()).cpu().numpy()
        want = np.float32(1.0) / x
        bad += int((got.view(np.uint32) != want.view(np.uint32)).sum())
        neg = ops.selftest_reciprocal(torch.from_numpy(-x).cuda()).cpu().numpy()
        bad += int((neg.view(np.uint32) != (-want).view(np.uint32)).sum())
    assert bad == 0
    # densities of a running simulation, both precisions
    rng = np.random.default_rng(1)
    for dt in (np.float32, np.float64):
        x = (1.0 + 0.3 * rng.standard_normal(1 << 20)).astype(dt)
        x = x[np.abs(x) > 1e-3]
        got = ops.selftest_reciprocal(torch.from_numpy(x).cuda()).cpu().numpy()
        assert np.array_equal(got, dt(1.0) / x)
    # fp64 (seven fused operations from the hardware estimate): sixteen million random significands of
    # one binade, the significands next to a power of two from either side, and far binades
    bad = 0
    for _ in range(4):
        x = 1.0 + rng.random(1 << 22)
        got = ops.selftest_reciprocal(torch.from_numpy(x).cuda()).cpu().numpy()
        bad += int((got.view(np.uint64) != (1.0 / x).view(np.uint64)).sum())
    edge = np.concatenate([np.uint64(0x3FF0000000000000) + np.arange(1 << 16, dtype=np.uint64),
                           np.uint64(0x3FFFFFFFFFFFFFFF) - np.arange(1 << 16, dtype=np.uint64)]).view(np.float64)
    got = ops.selftest_reciprocal(torch.from_numpy(edge).cuda()).cpu().numpy()
    bad += int((got.view(np.uint64) != (1.0 / edge).view(np.uint64)).sum())
    far = (1.0 + rng.random(1 << 18)) * 2.0 ** rng.integers(-1000, 1000, 1 << 18).astype(np.float64)
    far = np.concatenate([far, -far])
    got = ops.selftest_reciprocal(torch.from_numpy(far).cuda()).cpu().numpy()
    bad += int((got.view(np.uint64) != (1.0 / far).view(np.uint64)).sum())
    assert bad == 0
    # far binades (scale invariance, spot-checked)
    x = (np.float32(1.0) + rng.random(1 << 16, dtype=np.float32)) * np.float32(2.0) ** rng.integers(-120, 120, 1 << 16).astype(np.float32)
    got = ops.selftest_reciprocal(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(got, np.float32(1.0) / x)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scene,scheme", list(ONE_STEP))
def test_one_step_entry_points(dtype, scene, scheme):
    from lbm_amd import ops
    rng = np.random.default_rng(5)
    for nx, ny in ((64, 48), (229, 11), (3, 4), (1024, 70), (2, 2)):
        f = random_state(rng, nx, ny, dtype)
        d_f = torch.from_numpy(f).cuda()
        want = oracle_steps(ONE_STEP[(scene, scheme)], f, 1, 0.9, dtype)
        d_b = torch.full_like(d_f, float("nan"))
        if scene == "shear":
            getattr(ops, f"{scheme}_periodic_no_mpi")(d_f, d_b, nx, ny, 0.9)
        else:
            getattr(ops, f"{scheme}_lid_driven")(d_f, d_b, nx, ny, 0.9, RHO, U)
        assert_bits_equal(d_b.cpu().numpy(), want, f"contracted {scene}/{scheme} one step {nx}x{ny}")
        # not the strict arithmetic: last bits differ, nothing more
        with orc.arithmetic("unrolled"):
            strict = oracle_steps(ONE_STEP[(scene, scheme)], f, 1, 0.9, dtype)
        if nx * ny > 100:
            assert (strict != want).any()
        assert np.abs(strict.astype(np.float64) - want.astype(np.float64)).max() < (1e-6 if dtype == np.float32 else 1e-15)
        if scene == "shear" and ny >= 4:
            want = np.full_like(f, -7.0)
            getattr(orc, f"{scheme}_rows")(f, want, dtype(0.9), 1, ny - 1, False)
            d_b = torch.full_like(d_f, -7.0)
            getattr(ops, f"{scheme}_periodic_rows")(d_f, d_b, 0.9, 1, ny - 1, False)
            assert_bits_equal(d_b.cpu().numpy(), want, f"contracted {scheme} rows {nx}x{ny}")
    if scene == "shear":
        nx, ny = 70, 23
        f = random_state(rng, nx, ny, dtype)
        d_f = torch.from_numpy(f).cuda()
        want = np.zeros_like(f)
        getattr(orc, f"{scheme}_periodic_inner")(f, want, dtype(0.9), 2, 2, nx - 2, ny - 2)
        getattr(orc, f"{scheme}_periodic_outer")(f, want, dtype(0.9))
        d_b = torch.zeros_like(d_f)
        getattr(ops, f"{scheme}_periodic_inner")(d_f, d_b, 0.9, 2, 2, nx - 2, ny - 2)
        getattr(ops, f"{scheme}_periodic_outer")(d_f, d_b, 0.9, nx, ny)
        assert_bits_equal(d_b.cpu().numpy(), want, f"contracted {scheme} inner+outer")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scene,scheme", list(ONE_STEP))
@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6, 7, 8])
def test_several_steps_per_call(dtype, scene, scheme, k):
    from lbm_amd import ops
    rng = np.random.default_rng(60 + k)
    for nx, ny in SIZES:
        if scene == "lid" and (ny < 2 or nx < 2):
            continue
        f = random_state(rng, nx, ny, dtype)
        d_f = torch.from_numpy(f).cuda()
        if ops.fused_max_steps(d_f) == 1:
            continue   # fp32 rows of an odd length, rows of one node
        want = oracle_steps(ONE_STEP[(scene, scheme)], f, k, 1.3, dtype)
        d_b = torch.full_like(d_f, float("nan"))
        run_fused(scene, scheme, d_f, d_b, 1.3, 0, ny, k, wrap_y=True, y_goff=0, ny_glob=ny)
        assert_bits_equal(d_b.cpu().numpy(), want, f"contracted {scene}/{scheme} fused k={k} {nx}x{ny}")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scheme", ["pull", "push"])
@pytest.mark.parametrize("k", [7, 8])
def test_row_ranges_with_deep_ghost_rows_at_seven_and_eight_steps(dtype, scheme, k):
    rng = np.random.default_rng(300 + k)
    for nx, ny in ((256, 29), (1024, 40), (64, 2 * k + 1)):
        f = random_state(rng, nx, ny, dtype)
        want = oracle_steps(f"{scheme}_periodic_no_mpi", f, k, 0.8, dtype)
        d_f = torch.from_numpy(f).cuda()
        d_b = torch.full_like(d_f, -7.0)
        run_fused("shear", scheme, d_f, d_b, 0.8, k, ny - k, k, wrap_y=False)
        got = d_b.cpu().numpy()
        assert_bits_equal(got[:, k:ny - k], want[:, k:ny - k], f"{scheme} rows k={k} {nx}x{ny}")
        assert (got[:, :k] == -7.0).all() and (got[:, ny - k:] == -7.0).all()
    # a slab of a taller cavity, under the lid and at the bottom wall
    nx, ny_glob = 256, 48
    f = random_state(rng, nx, ny_glob, dtype)
    want = oracle_steps(f"{scheme}_lid_driven", f, k, 1.6, dtype)
    for g0, g1 in ((0, 17), (17, 31), (31, 48)):
        lo, hi = max(0, g0 - k), min(ny_glob, g1 + k)
        d_f = torch.from_numpy(np.ascontiguousarray(f[:, lo:hi])).cuda()
        d_b = torch.full_like(d_f, -7.0)
        run_fused("lid", scheme, d_f, d_b, 1.6, g0 - lo, g1 - lo, k, y_goff=lo, ny_glob=ny_glob)
        got = d_b.cpu().numpy()
        assert_bits_equal(got[:, g0 - lo:g1 - lo], want[:, g0:g1], f"lid {scheme} k={k} rows {g0}:{g1}")


def test_limits():
    from lbm_amd import capi, ops
    even = torch.zeros((9, 6, 64), dtype=torch.float32, device="cuda")
    assert ops.fused_max_steps(even) == 8
    assert ops.fused_max_steps(even.double()) == 8
    odd = torch.zeros((9, 6, 67), dtype=torch.float32, device="cuda")
    assert ops.fused_max_steps(odd) == 1
    with pytest.raises(capi.LbmError, match="unsupported"):
        ops.pull_periodic_fused(odd, torch.zeros_like(odd), 0.7, 0, 6, True, 2)
    with pytest.raises(capi.LbmError):
        ops.pull_periodic_fused(even, torch.zeros_like(even), 0.7, 0, 6, True, 9)
    with ops.layout("aos"):
        a = torch.zeros((6, 64, 9), dtype=torch.float32, device="cuda")
        assert ops.fused_max_steps(a) == 8
        a = torch.zeros((6, 67, 9), dtype=torch.float32, device="cuda")
        assert ops.fused_max_steps(a) == 1
        with pytest.raises(capi.LbmError, match="unsupported"):
            ops.pull_periodic_no_mpi(a, torch.zeros_like(a), 67, 6, 0.7)
    # a run is cut into equal launches of at most eight steps
    big = torch.zeros((9, 4096, 4096), dtype=torch.float32, device="cuda")
    assert ops.plan_steps(big, 4096, 100) <= 8
    assert ops.plan_steps(big, 4096, 20) == 7 and ops.plan_steps(big, 4096, 13) == 7 and ops.plan_steps(big, 4096, 6) == 6
    # contracted fp32: eight steps, whatever the scene
    assert ops.plan_steps(big, 4096, 800) == 8 and ops.plan_steps(big, 4096, 800, lid=True) == 8
    with ops.arithmetic("unrolled"):
        assert ops.plan_steps(big, 4096, 20) == 7 and ops.plan_steps(big, 4096, 500) == 7
        assert ops.plan_steps(big, 4096, 9) == 5 and ops.plan_steps(big, 4096, 3) == 3


@pytest.mark.parametrize("scene,scheme", list(ONE_STEP))
@pytest.mark.parametrize("tdt,ndt,budget", [(torch.float32, np.float32, 1e-5), (torch.float64, np.float64, 1e-12)])
def test_whole_runs_stay_inside_the_parity_budget_of_the_strict_spelling(scene, scheme, tdt, ndt, budget):
    """3000 steps on the device in the contracted arithmetic: bit-equal to the contracted oracle, and
    within 1e-5 (fp32) / 1e-12 (fp64) max-abs on populations of the strict oracle."""
    from lbm_amd import Simulation
    nx, ny, steps, omega = 128, 96, 3000, 1.7 if scene == "lid" else 0.5
    sim = Simulation(nx, ny, scene=scene, scheme=scheme, dtype=tdt, omega=omega)
    sim.step(steps)
    got = sim.f.cpu().numpy()
    want, _ = orc.run(scene, scheme, nx, ny, steps, omega, ndt)
    assert_bits_equal(got, want, f"contracted run {scene}/{scheme}")
    with orc.arithmetic("unrolled"):
        strict, _ = orc.run(scene, scheme, nx, ny, steps, omega, ndt)
    diff = np.abs(got.astype(np.float64) - strict.astype(np.float64)).max()
    assert 0 < diff < budget, diff
    assert sim.max_velocity() == float(orc.max_velocity(want))


@pytest.mark.parametrize("tdt,ndt,scheme", [(torch.float32, np.float32, "pull"), (torch.float32, np.float32, "push"),
                                            (torch.float64, np.float64, "pull"), (torch.float64, np.float64, "push")])
def test_full_size_shear_wave_equals_narrow_oracle(tdt, ndt, scheme):
    """16384 x 16384, 21 steps in launches of seven or eight (workgroups of four wavefronts where
    their rings fit): every column against the contracted oracle's 8-column run."""
    from lbm_amd import Simulation
    nx = ny = 16384
    sim = Simulation(nx, ny, scene="shear", scheme=scheme, dtype=tdt, omega=0.5)
    sim.step(21)
    assert sum(sim.launch_steps) == 21 and max(sim.launch_steps) > 6
    want, _ = orc.run("shear", scheme, 8, ny, 21, 0.5, ndt)
    col = torch.from_numpy(np.ascontiguousarray(want[:, :, :1])).cuda()
    bits = torch.int32 if tdt == torch.float32 else torch.int64
    assert all(bool((sim.f[q].view(bits) == col[q].view(bits)).all()) for q in range(9))
    del sim
    torch.cuda.empty_cache()


@pytest.mark.parametrize("idx", [0, 4])
def test_recorded_decay_series(golden_dir, idx):
    """The reference's recorded -omv series (1000 x 1000, fp32, 200 000 steps, docs/decays.js) with
    the contracted arithmetic, as shear-wave.py:60 runs it."""
    from lbm_amd import Simulation
    with open(os.path.join(golden_dir, "decays.json")) as fh:
        gold = json.load(fh)
    s = gold["series"][idx]
    sim = Simulation(gold["nx"], gold["ny"], scene="shear", scheme="pull", omega=s["omega"], dtype=torch.float32)
    text = sim.run(200000, gold["every"], "omv")
    got = np.array([float(v) for v in text.split(",")[:-1]])
    want = np.array(s["max_us"])
    assert got.shape == want.shape == (400,)
    # two correct fp32 evaluations of this run agree to ~4e-5 absolute (rounding noise of 200 000
    # steps in a maximum over 10^6 nodes; tests/test_oracle.py holds the oracle to the same numbers)
    assert np.abs(got - want).max() <= 5e-5
    np.testing.assert_allclose(got[:20], want[:20], rtol=1e-4)
