"""The reference's UNROLL_LOOPS=false arithmetic (pull_periodic_for / push_periodic_for,
pull-push.cpp:77-120, :190-243) in the periodic sweeps: bit for bit against the oracle's
restatement of those loops, and different from the default spelling in the last bits only."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from test_parity_gpu import assert_bits_equal
from test_fused_gpu import random_state

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def rolled_arithmetic():
    from lbm_amd import ops
    orc.set_rolled(True)
    try:
        with ops.unroll_loops(False):
            yield
    finally:
        orc.set_rolled(False)
    assert ops.get_unroll_loops() and not orc.get_rolled()


def oracle_steps(scheme, f, k, om, dtype):
    a, b = f.copy(), np.full_like(f, np.nan)
    for _ in range(k):
        getattr(orc, f"{scheme}_periodic_no_mpi")(a, b, dtype(om))
        a, b = b, a
    return a


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scheme", ["pull", "push"])
def test_one_step_entry_points(dtype, scheme):
    from lbm_amd import ops
    rng = np.random.default_rng(3)
    for nx, ny in ((64, 48), (229, 11), (3, 4), (1024, 70)):
        f = random_state(rng, nx, ny, dtype)
        d_f = torch.from_numpy(f).cuda()
        want = oracle_steps(scheme, f, 1, 0.9, dtype)
        d_b = torch.full_like(d_f, float("nan"))
        getattr(ops, f"{scheme}_periodic_no_mpi")(d_f, d_b, nx, ny, 0.9)
        assert_bits_equal(d_b.cpu().numpy(), want, f"rolled {scheme} no_mpi {nx}x{ny}")
        # not the default arithmetic: some last bits differ, nothing more
        orc.set_rolled(False)
        other = oracle_steps(scheme, f, 1, 0.9, dtype)
        orc.set_rolled(True)
        if nx * ny > 100:
            assert (other != want).any()
        assert np.abs(other.astype(np.float64) - want.astype(np.float64)).max() < (1e-6 if dtype == np.float32 else 1e-15)
        if ny >= 4:
            # row range with ghost rows, rows outside untouched
            want = np.full_like(f, -7.0)
            getattr(orc, f"{scheme}_rows")(f, want, dtype(0.9), 1, ny - 1, False)
            d_b = torch.full_like(d_f, -7.0)
            getattr(ops, f"{scheme}_periodic_rows")(d_f, d_b, 0.9, 1, ny - 1, False)
            assert_bits_equal(d_b.cpu().numpy(), want, f"rolled {scheme} rows {nx}x{ny}")
    # the halo'd forms (pull-push.cpp:303, :325, :347-416)
    nx, ny = 70, 23
    f = random_state(rng, nx, ny, dtype)
    d_f = torch.from_numpy(f).cuda()
    want = np.zeros_like(f)
    getattr(orc, f"{scheme}_periodic_inner")(f, want, dtype(0.9), 2, 2, nx - 2, ny - 2)
    getattr(orc, f"{scheme}_periodic_outer")(f, want, dtype(0.9))
    d_b = torch.zeros_like(d_f)
    getattr(ops, f"{scheme}_periodic_inner")(d_f, d_b, 0.9, 2, 2, nx - 2, ny - 2)
    getattr(ops, f"{scheme}_periodic_outer")(d_f, d_b, 0.9, nx, ny)
    assert_bits_equal(d_b.cpu().numpy(), want, f"rolled {scheme} inner+outer")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scheme", ["pull", "push"])
@pytest.mark.parametrize("k", [1, 2, 4, 6, 7, 8])
def test_several_steps_per_call(dtype, scheme, k):
    from lbm_amd import ops
    rng = np.random.default_rng(40 + k)
    for nx, ny in ((64, 48), (112, 9), (1024, 70), (8, 3), (724, 33)):
        f = random_state(rng, nx, ny, dtype)
        want = oracle_steps(scheme, f, k, 1.3, dtype)
        d_f = torch.from_numpy(f).cuda()
        d_b = torch.full_like(d_f, float("nan"))
        getattr(ops, f"{scheme}_periodic_fused")(d_f, d_b, 1.3, 0, ny, True, k)
        assert_bits_equal(d_b.cpu().numpy(), want, f"rolled {scheme} fused k={k} {nx}x{ny}")


def test_limits_and_what_ignores_the_switch():
    from lbm_amd import capi, ops
    odd = torch.zeros((9, 6, 67), dtype=torch.float32, device="cuda")
    assert ops.fused_max_steps(odd) == 1                       # fp32 rows of an odd length: one step per call
    assert ops.fused_max_steps(torch.zeros((9, 6, 67), dtype=torch.float64, device="cuda")) == 8
    with pytest.raises(capi.LbmError, match="unsupported"):
        ops.pull_periodic_fused(odd, torch.zeros_like(odd), 0.7, 0, 6, True, 2)
    with ops.layout("aos"):
        a = torch.zeros((6, 64, 9), dtype=torch.float32, device="cuda")
        with pytest.raises(capi.LbmError, match="unsupported"):
            ops.pull_periodic_no_mpi(a, torch.zeros_like(a), 64, 6, 0.7)
    # the cavity has one spelling in the reference (pull-push.cpp:432-531)
    rng = np.random.default_rng(9)
    f = random_state(rng, 64, 20, np.float32)
    want = np.zeros_like(f)
    orc.push_lid_driven(f, want, np.float32(1.1), np.float32(1.0), np.float32(0.1))
    orc.set_rolled(False)
    same = np.zeros_like(f)
    orc.push_lid_driven(f, same, np.float32(1.1), np.float32(1.0), np.float32(0.1))
    orc.set_rolled(True)
    assert (want == same).all()
    d_f = torch.from_numpy(f).cuda()
    d_b = torch.zeros_like(d_f)
    ops.push_lid_driven(d_f, d_b, 64, 20, 1.1, 1.0, 0.1)
    assert_bits_equal(d_b.cpu().numpy(), want, "lid push under the rolled switch")


def test_simulation_and_decay_rate():
    """A whole run in the rolled arithmetic: equal to the oracle's, and the shear wave decays at the
    analytic rate (shear-wave.py:55,78)."""
    from lbm_amd.sim import Simulation
    sim = Simulation(128, 96, scene="shear", scheme="pull", omega=1.0)
    assert sim.max_velocity() == pytest.approx(0.1, rel=1e-6)
    sim.step(500)
    want, _ = orc.run("shear", "pull", 128, 96, 500, 1.0, np.float32)
    assert_bits_equal(sim.f.cpu().numpy(), want, "rolled simulation")
    nu = (1.0 / 1.0 - 0.5) / 3.0
    assert sim.max_velocity() == pytest.approx(0.1 * np.exp(-nu * (2 * np.pi / 96) ** 2 * 500), rel=1e-2)
