"""CPU stand-in for the slab kernels, built on the oracle: lets the y-slab exchange schedule of
lbm_amd.slab run on CPU ranks (gloo) in tests.  Test infrastructure only."""
import numpy as np

from oracle import oracle as orc


class OracleBackend:
    def init_shearwave(self, f, rho, u, y_below, y_glob):
        a = f.numpy()
        interior = np.ascontiguousarray(a[:, 1:-1])
        t = a.dtype.type
        orc.init_shearwave(interior, t(rho), t(u), y_below, y_glob)
        a[:, 1:-1] = interior

    def init_rest(self, f, rho):
        a = f.numpy()
        orc.init_rest(a, a.dtype.type(rho))

    def pull_periodic_rows(self, f, buf, om, y_lo, y_hi):
        a = f.numpy()
        orc.pull_rows(a, buf.numpy(), a.dtype.type(om), y_lo, y_hi, False)

    def push_periodic_rows(self, f, buf, om, y_lo, y_hi):
        a = f.numpy()
        orc.push_rows(a, buf.numpy(), a.dtype.type(om), y_lo, y_hi, False)

    def pull_lid_rows(self, f, buf, om, rho, u, y_lo, y_hi, y_goff, ny_glob):
        a = f.numpy(); t = a.dtype.type
        orc.pull_lid_rows(a, buf.numpy(), t(om), t(rho), t(u), y_lo, y_hi, y_goff, ny_glob)

    def push_lid_rows(self, f, buf, om, rho, u, y_lo, y_hi, y_goff, ny_glob):
        a = f.numpy(); t = a.dtype.type
        orc.push_lid_rows(a, buf.numpy(), t(om), t(rho), t(u), y_lo, y_hi, y_goff, ny_glob)

    def max_velocity_dev(self, out, f, y_lo, y_hi):
        out[0] = float(orc.max_velocity(f.numpy(), y_lo, y_hi))

    def compute_velocities(self, vel, f, y_lo, y_hi):
        vel.numpy()[:] = orc.compute_velocities(f.numpy(), y_lo, y_hi)
