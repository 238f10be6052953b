"""CPU stand-in for the slab kernels, built on the oracle: lets the y-slab exchange schedule of
lbm_amd.slab run on CPU ranks (gloo) in tests.  Test infrastructure only."""
import numpy as np

from oracle import oracle as orc


class OracleBackend:
    # steps per launch that SlabSimulation(halo=None) would be told to take; tests set it to pick
    # the one-row (1) or the deep-halo (> 1) schedule
    fused_depth = 1

    def init_shearwave(self, f, rho, u, y_below, y_glob, halo=1):
        a = f.numpy()
        interior = np.ascontiguousarray(a[:, halo:a.shape[1] - halo])
        t = a.dtype.type
        orc.init_shearwave(interior, t(rho), t(u), y_below, y_glob)
        a[:, halo:a.shape[1] - halo] = interior

    def plan_steps(self, f, rows, remaining, lid=False):
        return min(self.fused_depth, remaining)

    # k steps of rows [y_lo, y_hi) from an array with >= k ghost rows: step j advances rows
    # [y_lo-(k-j), y_hi+(k-j)), one oracle sweep at a time.  The push sweep scatters, so it is run
    # on one more row at either end (inside a copy padded by one row) and only complete rows are kept.
    def _fused(self, one_step, f, buf, y_lo, y_hi, k, clip_lo, clip_hi):
        a = f.numpy()
        ny = a.shape[1]
        cur = np.zeros((9, ny + 2, a.shape[2]), dtype=a.dtype)
        cur[:, 1:ny + 1] = a
        for j in range(1, k + 1):
            lo = max(y_lo - (k - j), clip_lo)
            hi = min(y_hi + (k - j), clip_hi)
            nxt = np.zeros_like(cur)
            one_step(cur, nxt, lo + 1, hi + 1)      # +1: row index inside the padded copy
            keep = np.zeros_like(cur)
            keep[:, lo + 1:hi + 1] = nxt[:, lo + 1:hi + 1]
            cur = keep
        buf.numpy()[:, y_lo:y_hi] = cur[:, y_lo + 1:y_hi + 1]

    def periodic_fused(self, scheme, f, buf, om, y_lo, y_hi, k):
        t = f.numpy().dtype.type
        ny = f.shape[1]
        if scheme == "pull":
            step = lambda a, b, lo, hi: orc.pull_rows(a, b, t(om), lo, hi, False)
        else:
            step = lambda a, b, lo, hi: orc.push_rows(a, b, t(om), max(lo - 1, 1), min(hi + 1, ny + 1), False)
        self._fused(step, f, buf, y_lo, y_hi, k, 0, ny)

    def lid_fused(self, scheme, f, buf, om, rho, u, y_lo, y_hi, y_goff, ny_glob, k):
        t = f.numpy().dtype.type
        ny = f.shape[1]
        # rows of the local array that lie inside the cavity
        clip_lo, clip_hi = max(0, -y_goff), min(ny, ny_glob - y_goff)
        goff = y_goff - 1   # global row of padded row 0
        if scheme == "pull":
            step = lambda a, b, lo, hi: orc.pull_lid_rows(a, b, t(om), t(rho), t(u), lo, hi, goff, ny_glob)
        else:
            step = lambda a, b, lo, hi: orc.push_lid_rows(a, b, t(om), t(rho), t(u), max(lo - 1, clip_lo + 1),
                                                          min(hi + 1, clip_hi + 1), goff, ny_glob)
        self._fused(step, f, buf, y_lo, y_hi, k, clip_lo, clip_hi)

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


class OracleCartBackend(OracleBackend):
    """CPU stand-in for the block kernels of lbm_amd.cart, built on the oracle: inner / outer sweeps
    for the halo ring of one node, and the multi-step sweep of the slab stand-in for deep frames
    (fused_depth > 1 makes CartSimulation(halo=None) choose them)."""

    def pull_inner(self, f, buf, om, LX, LY, HX, HY):
        a = f.numpy()
        orc.pull_periodic_inner(a, buf.numpy(), a.dtype.type(om), LX, LY, HX, HY)

    def pull_outer(self, f, buf, om):
        a = f.numpy()
        orc.pull_periodic_outer(a, buf.numpy(), a.dtype.type(om))

    def push_inner(self, f, buf, om, LX, LY, HX, HY):
        a = f.numpy()
        orc.push_periodic_inner(a, buf.numpy(), a.dtype.type(om), LX, LY, HX, HY)

    def push_outer(self, f, buf, om):
        a = f.numpy()
        orc.push_periodic_outer(a, buf.numpy(), a.dtype.type(om))

    def compute_velocities(self, vel, f, y_lo, y_hi):
        vel.numpy()[...] = orc.compute_velocities(f.numpy(), y_lo, y_hi)
