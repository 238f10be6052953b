"""lbm_max_velocity_dev_static: the max-|u| reduction for callers that cannot allocate device memory.

Called through the C ABI and compared bit for bit with lbm_max_velocity_dev (the entry point it is a
variant of) on two calls in a row, the second on a smaller lattice, so that anything the first call
leaves in the per-device static scratch array would show.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _state(nx, ny, dtype, seed):
    """Random populations near equilibrium: positive densities, velocities of a few per cent."""
    g = torch.Generator().manual_seed(seed)
    w = torch.tensor([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=torch.float64).view(9, 1, 1)
    f = w * (1.0 + 0.2 * (torch.rand((9, ny, nx), generator=g, dtype=torch.float64) - 0.5))
    return f.to(dtype).cuda().contiguous()


def _both(f, y_lo, y_hi):
    from lbm_amd import capi
    prec = f.element_size()
    _, ny, nx = f.shape
    stream = torch.cuda.current_stream().cuda_stream
    a = torch.full((2,), -7.0, dtype=f.dtype, device=f.device)
    b = torch.full((2,), -9.0, dtype=f.dtype, device=f.device)
    capi.call("lbm_max_velocity_dev", f.data_ptr(), a.data_ptr(), prec, nx, ny, y_lo, y_hi, stream)
    capi.call("lbm_max_velocity_dev_static", f.data_ptr(), b.data_ptr(), prec, nx, ny, y_lo, y_hi, stream)
    torch.cuda.synchronize()
    return a[:1].cpu().numpy(), b[:1].cpu().numpy()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_static_scratch_reduction_equals_the_allocating_one_twice_in_a_row(dtype):
    from oracle import oracle as orc
    big = _state(1024, 384, dtype, 1)
    small = _state(96, 40, dtype, 2) * 0.999          # lower speeds: a stale key would win
    small[1] += 1e-4
    a, b = _both(big, 0, 384)
    assert a.tobytes() == b.tobytes()
    assert b[0] == orc.max_velocity(big.cpu().numpy())
    free0, _ = torch.cuda.mem_get_info()
    a2, b2 = _both(small, 3, 37)                       # smaller lattice, a row range
    free1, _ = torch.cuda.mem_get_info()
    assert a2.tobytes() == b2.tobytes()
    assert b2[0] == orc.max_velocity(small.cpu().numpy(), 3, 37)
    assert b2[0] != b[0]
    assert free1 == free0                              # no device memory taken by the second round
    # an empty row range starts from the reference's -1e30 (io.cpp:184), whatever the array held
    a3, b3 = _both(small, 5, 5)
    assert a3.tobytes() == b3.tobytes() and b3[0] == np.dtype(a3.dtype).type(-1e30)


def test_static_scratch_rejects_what_the_abi_rejects():
    from lbm_amd import capi
    f = _state(64, 16, torch.float32, 3)
    out = torch.zeros(2, dtype=torch.float32, device="cuda")
    with pytest.raises(capi.LbmError):
        capi.call("lbm_max_velocity_dev_static", f.data_ptr(), out.data_ptr(), 4, 1 << 31, 16, 0, 16, None)
    with pytest.raises(capi.LbmError):
        capi.call("lbm_max_velocity_dev_static", f.data_ptr(), out.data_ptr() + 4, 4, 64, 16, 0, 16, None)
    with pytest.raises(capi.LbmError):
        capi.call("lbm_max_velocity_dev_static", f.data_ptr(), out.data_ptr(), 4, 64, 16, 4, 17, None)
