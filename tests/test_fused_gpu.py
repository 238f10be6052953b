"""Several time steps per call (lbm_*_fused) against the CPU oracle stepped one step at a time.

Bit-exact: the fused kernels run the same per-node arithmetic in the same order as the one-step
sweeps, only the intermediate time levels stay in registers.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from test_parity_gpu import assert_bits_equal, TORCH

pytestmark = pytest.mark.gpu

RHO, U = 1.0, 0.1


def random_state(rng, nx, ny, dtype):
    return (1.0 / 9 + 0.02 * rng.standard_normal((9, ny, nx))).astype(dtype)


def oracle_steps(name, f, k, om, dtype):
    """k one-step oracle sweeps of a whole lattice with the buffers swapped in between."""
    a = f.copy()
    b = np.full_like(f, np.nan)
    for _ in range(k):
        if "lid" in name:
            getattr(orc, name)(a, b, dtype(om), dtype(RHO), dtype(U))
        else:
            getattr(orc, name)(a, b, dtype(om))
        a, b = b, a
    return a


ONE_STEP = {("shear", "pull"): "pull_periodic_no_mpi", ("shear", "push"): "push_periodic_no_mpi",
            ("lid", "pull"): "pull_lid_driven", ("lid", "push"): "push_lid_driven"}


def run_fused(scene, scheme, d_f, d_b, om, y_lo, y_hi, k, wrap_y=True, y_goff=0, ny_glob=None):
    from lbm_amd import ops
    if scene == "shear":
        getattr(ops, f"{scheme}_periodic_fused")(d_f, d_b, om, y_lo, y_hi, wrap_y, k)
    else:
        getattr(ops, f"{scheme}_lid_fused")(d_f, d_b, om, RHO, U, y_lo, y_hi, y_goff, ny_glob, k)


# row lengths: one strip, a partial last strip, several strips (a strip stores 56 lanes = 112 fp32 /
# 56 fp64 nodes; 48 lanes for odd fp32 rows and for fp64 beyond four steps), the shortest supported
# rows, odd lengths; heights: fewer rows than steps (multiple wraps), one segment, several segments
SIZES = [(8, 3), (8, 1), (64, 48), (112, 9), (114, 7), (1024, 70), (2048, 5), (4, 6), (724, 33), (2, 5), (57, 6),
         (229, 11), (3, 4), (1027, 9), (8, 2), (2, 2), (113, 2)]   # the last three: lid and bottom wall on adjacent rows


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scene,scheme", list(ONE_STEP))
@pytest.mark.parametrize("k", [2, 3, 4, 5, 6, 7, 8])
def test_whole_lattice_matches_k_oracle_steps(dtype, scene, scheme, k):
    from lbm_amd import ops
    rng = np.random.default_rng(7 * k + (dtype == np.float64))
    for nx, ny in SIZES:
        if scene == "lid" and (ny < 2 or nx < 2):
            continue
        f = random_state(rng, nx, ny, dtype)
        om = 1.3
        want = oracle_steps(ONE_STEP[(scene, scheme)], f, k, om, dtype)
        d_f = torch.from_numpy(f).cuda()
        d_b = torch.full_like(d_f, float("nan"))
        if ops.fused_max_steps(d_f) == 1:
            continue   # rows of a single node
        run_fused(scene, scheme, d_f, d_b, om, 0, ny, k, wrap_y=True, y_goff=0, ny_glob=ny)
        assert_bits_equal(d_b.cpu().numpy(), want, f"{scene}/{scheme} fused k={k} {nx}x{ny}")
        assert torch.equal(d_f.cpu(), torch.from_numpy(f)), "the input array was modified"


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scheme", ["pull", "push"])
@pytest.mark.parametrize("k", [2, 3, 4, 6, 7, 8])
def test_periodic_row_range_with_deep_ghost_rows(dtype, scheme, k):
    """wrap_y = 0: rows [k, ny-k) advance from an array whose outer k rows are ghost rows; rows
    outside the range must not be written."""
    rng = np.random.default_rng(100 + k)
    for nx, ny in ((256, 21), (1024, 40), (64, 2 * k + 1)):
        f = random_state(rng, nx, ny, dtype)
        want = oracle_steps(f"{scheme}_periodic_no_mpi", f, k, 0.8, dtype)
        d_f = torch.from_numpy(f).cuda()
        d_b = torch.full_like(d_f, -7.0)
        run_fused("shear", scheme, d_f, d_b, 0.8, k, ny - k, k, wrap_y=False)
        got = d_b.cpu().numpy()
        assert_bits_equal(got[:, k:ny - k], want[:, k:ny - k], f"{scheme} rows k={k} {nx}x{ny}")
        assert (got[:, :k] == -7.0).all() and (got[:, ny - k:] == -7.0).all()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scheme", ["pull", "push"])
@pytest.mark.parametrize("k", [2, 4, 5, 6, 7, 8])
def test_lid_slab_of_a_taller_cavity(dtype, scheme, k):
    """A slab holds global rows [g0 - ghosts, g1 + ghosts) of a cavity of height 40; slabs at the
    bottom wall, in the middle and under the lid."""
    rng = np.random.default_rng(200 + k)
    nx, ny_glob = 256, 40
    f = random_state(rng, nx, ny_glob, dtype)
    want = oracle_steps(f"{scheme}_lid_driven", f, k, 1.6, dtype)
    for g0, g1 in ((0, 13), (13, 27), (27, 40), (0, 40)):
        lo = max(0, g0 - k)          # first global row present in the local array
        hi = min(ny_glob, g1 + k)
        local = np.ascontiguousarray(f[:, lo:hi])
        d_f = torch.from_numpy(local).cuda()
        d_b = torch.full_like(d_f, -7.0)
        run_fused("lid", scheme, d_f, d_b, 1.6, g0 - lo, g1 - lo, k, y_goff=lo, ny_glob=ny_glob)
        got = d_b.cpu().numpy()
        assert_bits_equal(got[:, g0 - lo:g1 - lo], want[:, g0:g1], f"lid {scheme} k={k} rows {g0}:{g1}")
        assert (got[:, :g0 - lo] == -7.0).all() and (got[:, g1 - lo:] == -7.0).all()


def test_one_step_is_forwarded_and_bad_arguments_are_rejected():
    from lbm_amd import capi, ops
    rng = np.random.default_rng(3)
    f = random_state(rng, 64, 12, np.float32)
    d_f = torch.from_numpy(f).cuda()
    d_b = torch.zeros_like(d_f)
    ops.pull_periodic_fused(d_f, d_b, 0.7, 0, 12, True, 1)
    want = np.zeros_like(f)
    orc.pull_periodic_no_mpi(f, want, np.float32(0.7))
    assert_bits_equal(d_b.cpu().numpy(), want, "nsteps=1")
    assert ops.fused_max_steps(d_f) == 8
    odd = torch.zeros((9, 4, 67), dtype=torch.float64, device="cuda")
    assert ops.fused_max_steps(odd) == 8      # one fp64 node per lane: any row length
    odd = torch.zeros((9, 4, 66), dtype=torch.float32, device="cuda")
    assert ops.fused_max_steps(odd) == 8
    odd = torch.zeros((9, 4, 67), dtype=torch.float32, device="cuda")
    assert ops.fused_max_steps(odd) == 8      # odd rows in fp32: one node per lane
    one = torch.zeros((9, 4, 1), dtype=torch.float32, device="cuda")
    assert ops.fused_max_steps(one) == 1
    with pytest.raises(capi.LbmError):
        ops.pull_periodic_fused(one, torch.zeros_like(one), 0.7, 0, 4, True, 2)     # unsupported row length
    with pytest.raises(capi.LbmError):
        ops.pull_periodic_fused(d_f, d_b, 0.7, 0, 12, True, 9)                      # too many steps
    with pytest.raises(capi.LbmError):
        ops.push_periodic_fused(d_f, d_b, 0.7, 1, 11, False, 2)                     # ghost rows too shallow
    with pytest.raises(capi.LbmError):
        ops.push_lid_fused(d_f, d_b, 0.7, 1.0, 0.1, 1, 11, 5, 40, 2)                # ghost rows too shallow
    with pytest.raises(capi.LbmError):
        ops.pull_periodic_fused(d_f, d_f, 0.7, 0, 12, True, 2)                      # overlap


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_extreme_aspect_ratios_fused_equals_one_step(dtype):
    """Very long rows (thousands of strips) and very tall, narrow lattices (thousands of segments):
    six and eight fused steps against as many launches of the one-step sweeps, which the oracle tests pin."""
    from lbm_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(11)
    for nx, ny in ((262144, 20), (6, 50000), (131070, 7)):
        f = 1.0 / 9 + 0.02 * torch.randn((9, ny, nx), dtype=dtype, device="cuda", generator=gen)
        for scene in ("shear", "lid"):
            for scheme in ("pull", "push"):
                for k in (6, 8):
                    a, b = f.clone(), torch.empty_like(f)
                    for _ in range(k):
                        if scene == "shear":
                            getattr(ops, f"{scheme}_periodic_no_mpi")(a, b, nx, ny, 1.2)
                        else:
                            getattr(ops, f"{scheme}_lid_driven")(a, b, nx, ny, 1.2, 1.0, 0.1)
                        a, b = b, a
                    got = torch.full_like(f, float("nan"))
                    if scene == "shear":
                        getattr(ops, f"{scheme}_periodic_fused")(f, got, 1.2, 0, ny, True, k)
                    else:
                        getattr(ops, f"{scheme}_lid_fused")(f, got, 1.2, 1.0, 0.1, 0, ny, 0, ny, k)
                    bits = torch.int32 if dtype == torch.float32 else torch.int64
                    assert bool((got.view(bits) == a.view(bits)).all()), (scene, scheme, nx, ny, k)


@pytest.fixture
def grouped_launches():
    """Workgroups of several wavefronts on lattices of any size; without the switch only lattices of
    many rounds of wavefronts are launched that way."""
    from lbm_amd import ops
    with ops.tuning("fused_group", 1):
        yield


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scene,scheme", list(ONE_STEP))
@pytest.mark.parametrize("k", [2, 4, 5, 7, 8])
@pytest.mark.parametrize("arith", ["unrolled", "contracted"])
def test_workgroups_of_several_wavefronts(grouped_launches, dtype, scene, scheme, k, arith):
    """The launch form of large lattices -- four adjacent strips per workgroup, rings in dynamic LDS
    -- on small and ragged ones: strips that do not fill their last group, one strip only, several
    segments, walls inside a group."""
    from lbm_amd import ops
    rng = np.random.default_rng(11 * k + (dtype == np.float64))
    with ops.arithmetic(arith), orc.arithmetic(arith):
        for nx, ny in [(64, 48), (1024, 70), (724, 33), (2048, 5), (1536, 300), (8, 3)]:
            f = random_state(rng, nx, ny, dtype)
            want = oracle_steps(ONE_STEP[(scene, scheme)], f, k, 0.9, dtype)
            d_f = torch.from_numpy(f).cuda()
            d_b = torch.full_like(d_f, float("nan"))
            run_fused(scene, scheme, d_f, d_b, 0.9, 0, ny, k, ny_glob=ny)
            assert_bits_equal(d_b.cpu().numpy(), want, f"{scene} {scheme} {arith} {nx}x{ny} k={k} grouped")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scene,scheme", list(ONE_STEP))
@pytest.mark.parametrize("threads_x", [64, 128, 256, 512])
def test_launch_shape_hint_sets_the_workgroup_width(dtype, scene, scheme, threads_x):
    """threads_x of the launch shape (the -tx tile width of the command line) is the workgroup width of
    the multi-step sweeps too: one, two or four wavefronts with adjacent strips.  Same bits in every
    width, on lattices whose strips do not fill the last workgroup."""
    from lbm_amd import ops
    rng = np.random.default_rng(threads_x + (dtype == np.float64))
    with ops.Context(threads_x=threads_x):
        for k in (3, 6, 8):
            for nx, ny in [(724, 33), (1536, 200), (112, 9)]:
                f = random_state(rng, nx, ny, dtype)
                want = oracle_steps(ONE_STEP[(scene, scheme)], f, k, 1.1, dtype)
                d_f = torch.from_numpy(f).cuda()
                d_b = torch.full_like(d_f, float("nan"))
                run_fused(scene, scheme, d_f, d_b, 1.1, 0, ny, k, ny_glob=ny)
                assert_bits_equal(d_b.cpu().numpy(), want, f"{scene} {scheme} {nx}x{ny} k={k} threads_x={threads_x}")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scene,scheme", [("shear", "pull"), ("shear", "push")])
@pytest.mark.parametrize("narrow", ["0", "1", "2"])
def test_five_steps_with_either_halo_width(dtype, scene, scheme, narrow):
    """Small periodic lattices take launches of five steps with halos of whole lanes (58 stored lanes
    per strip in fp32, 54 in fp64) where that makes fewer strips, everything else with halos of whole
    32-byte granules (56 / 48); the switch fused_narrow = 0 never takes narrow halos, 2 always.  Row
    lengths around the strip widths of both."""
    from lbm_amd import ops
    rng = np.random.default_rng(5 + (dtype == np.float64))
    with ops.tuning("fused_narrow", int(narrow)):
        for nx, ny in [(116, 9), (118, 9), (112, 7), (108, 7), (54, 11), (56, 11), (1024, 40), (1044, 13), (522, 64)]:
            f = random_state(rng, nx, ny, dtype)
            want = oracle_steps(ONE_STEP[(scene, scheme)], f, 5, 1.1, dtype)
            d_f = torch.from_numpy(f).cuda()
            d_b = torch.full_like(d_f, float("nan"))
            run_fused(scene, scheme, d_f, d_b, 1.1, 0, ny, 5, ny_glob=ny)
            assert_bits_equal(d_b.cpu().numpy(), want, f"{scene} {scheme} {nx}x{ny} five steps, narrow={narrow}")


@pytest.fixture
def bulk_switch(bulk):
    from lbm_amd import ops
    with ops.tuning("fused_bulk", int(bulk)):
        yield


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("scheme", ["pull", "push"])
@pytest.mark.parametrize("k", [3, 5, 7, 8])
@pytest.mark.parametrize("bulk", ["0", "1"])
def test_cavity_with_and_without_the_periodic_code_in_its_bulk(bulk_switch, dtype, scheme, k, bulk):
    """Cavities wide and high enough to have wavefronts k nodes or more from every wall: those run the
    periodic sweep's code (switch fused_bulk = 1, the default) or, like everybody else, the code with
    the wall cases (0).  Both must give the oracle's populations; whole cavities and a row range with
    a global offset (a slab of a higher cavity)."""
    rng = np.random.default_rng(3 * k + (dtype == np.float64))
    name = ONE_STEP[("lid", scheme)]
    for nx, ny in [(1536, 300), (700, 420)]:
        f = random_state(rng, nx, ny, dtype)
        want = oracle_steps(name, f, k, 1.3, dtype)
        d_f = torch.from_numpy(f).cuda()
        d_b = torch.full_like(d_f, float("nan"))
        run_fused("lid", scheme, d_f, d_b, 1.3, 0, ny, k, ny_glob=ny)
        assert_bits_equal(d_b.cpu().numpy(), want, f"lid {scheme} {nx}x{ny} k={k} bulk={bulk}")
        # rows [100, 260) of the same cavity, held in an array that starts at global row 100 - k
        lo, hi = 100 - k, 260 + k
        d_s = torch.from_numpy(np.ascontiguousarray(f[:, lo:hi])).cuda()
        d_o = torch.full_like(d_s, -7.0)
        run_fused("lid", scheme, d_s, d_o, 1.3, k, k + 160, k, y_goff=lo, ny_glob=ny)
        assert_bits_equal(d_o.cpu().numpy()[:, k:k + 160], want[:, 100:260], f"lid {scheme} {nx}x{ny} k={k} rows 100:260 bulk={bulk}")
