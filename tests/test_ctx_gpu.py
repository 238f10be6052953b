"""Settings that travel with an object (lbm_ctx / ops.Context): simulations with different layout
and arithmetic interleaved in one process, each bit-identical to the oracle in its own arithmetic,
whatever the process-wide settings are meanwhile; and two host threads with different contexts."""
import threading

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from test_parity_gpu import assert_bits_equal

pytestmark = pytest.mark.gpu


def populations(sim):
    f = sim.f
    if sim.ctx.layout == "aos":
        f = f.permute(2, 0, 1)
    return f.contiguous().cpu().numpy()


def oracle_state(arith, scene, scheme, nx, ny, steps, omega, dtype):
    with orc.arithmetic("unrolled" if arith == "strict" else arith):
        return orc.run(scene, scheme, nx, ny, steps, omega, dtype)[0]


def test_simulations_with_different_settings_interleaved():
    from lbm_amd import Simulation, ops
    cases = [
        dict(arith="contracted", layout="aos", scene="shear", scheme="pull", nx=96, ny=80, dtype=np.float32, omega=0.7),
        dict(arith="strict", layout="soa", scene="lid", scheme="push", nx=112, ny=72, dtype=np.float64, omega=1.3),
        dict(arith="rolled", layout="soa", scene="shear", scheme="push", nx=128, ny=64, dtype=np.float32, omega=1.0),
        dict(arith="contracted", layout="soa", scene="lid", scheme="pull", nx=100, ny=90, dtype=np.float32, omega=1.7),
        dict(arith="strict", layout="aos", scene="shear", scheme="push", nx=64, ny=48, dtype=np.float64, omega=0.5),
    ]
    assert ops.get_arithmetic() == "unrolled" and ops.get_layout() == "soa"
    sims = [Simulation(c["nx"], c["ny"], scene=c["scene"], scheme=c["scheme"],
                       dtype=torch.float32 if c["dtype"] == np.float32 else torch.float64, omega=c["omega"],
                       layout=c["layout"], arithmetic=c["arith"]) for c in cases]
    done = 0
    for chunk in (1, 7, 12, 30):
        # the process-wide settings change under the objects' feet: they must not care
        ops.set_arithmetic("contracted" if chunk % 2 else "unrolled")
        for sim in sims:
            sim.step(chunk)
        done += chunk
        for sim, c in zip(sims, cases):
            want = oracle_state(c["arith"], c["scene"], c["scheme"], c["nx"], c["ny"], done, c["omega"], c["dtype"])
            assert_bits_equal(populations(sim), want, f"{c} after {done} steps")
    ops.set_arithmetic("unrolled")
    # outputs run under the object's settings as well
    for sim, c in zip(sims, cases):
        want = oracle_state(c["arith"], c["scene"], c["scheme"], c["nx"], c["ny"], done, c["omega"], c["dtype"])
        assert sim.max_velocity() == float(orc.max_velocity(want))
    assert ops.get_arithmetic() == "unrolled" and ops.get_layout() == "soa"
    assert sims[0].ctx.layout == "aos" and sims[0].ctx.arithmetic == "contracted"


def test_settings_in_force_are_captured_at_construction():
    from lbm_amd import Simulation, ops
    with ops.arithmetic("contracted"):
        sim = Simulation(64, 48, omega=0.9)
    assert ops.get_arithmetic() == "unrolled" and sim.ctx.arithmetic == "contracted"
    sim.step(25)
    assert_bits_equal(populations(sim), oracle_state("contracted", "shear", "pull", 64, 48, 25, 0.9, np.float32), "captured")
    with ops.Context(arithmetic="contracted", layout="aos"):
        assert ops.get_arithmetic() == "contracted" and ops.get_layout() == "aos"
        inner = Simulation(64, 48, omega=0.9)
        assert tuple(inner.f.shape) == (48, 64, 9)
    assert ops.get_layout() == "soa"
    inner.step(25)
    assert_bits_equal(populations(inner), oracle_state("contracted", "shear", "pull", 64, 48, 25, 0.9, np.float32), "nested")


def test_two_host_threads_with_different_contexts():
    """A context is current per host thread: one thread runs the strict arithmetic, the other the
    contracted one, at the same time, through the plain operator functions."""
    from lbm_amd import ops
    nx, ny, steps, omega = 256, 128, 40, 1.1
    results, errors = {}, []

    def worker(arith):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream), ops.Context(arithmetic=arith):
                f = torch.empty((9, ny, nx), dtype=torch.float32, device="cuda")
                buf = torch.empty_like(f)
                ops.init_shearwave(f, 1.0, 0.1, 0, ny)
                for _ in range(steps):
                    ops.pull_periodic_no_mpi(f, buf, nx, ny, omega)
                    f, buf = buf, f
                stream.synchronize()
                results[arith] = f.cpu().numpy()
        except Exception as exc:  # surfaced in the main thread
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(a,)) for a in ("strict", "contracted")]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for arith in ("strict", "contracted"):
        assert_bits_equal(results[arith], oracle_state(arith, "shear", "pull", nx, ny, steps, omega, np.float32), arith)
    assert not np.array_equal(results["strict"], results["contracted"])


def test_kernels_load_ahead_of_the_first_launch():
    """lbm_load_kernels: every family of sweeps that a combination of settings selects is in the library
    (a missing instantiation would fail here, not at somebody's first launch)."""
    from lbm_amd import capi, ops
    for layout, arith in (("soa", "strict"), ("soa", "contracted"), ("soa", "rolled"), ("aos", "strict"), ("aos", "contracted")):
        with ops.Context(layout=layout, arithmetic=arith):
            capi.call("lbm_load_kernels", 4)
            capi.call("lbm_load_kernels", 8)
