/*
 * lbm_hip.h -- C ABI of the MI355X (gfx950) D2Q9 BGK lattice-Boltzmann hot path.
 *
 * Every entry point replaces one free function of the reference's operator layer
 * (reference: src/init.h, src/pull-push.h, src/io.h of JulianKarrer/lbm); the citation next to each
 * declaration names it.  Where the reference passes Kokkos View handles, this ABI takes the raw
 * device pointer of the same storage: populations are SoA [Q=9][NY][NX], x contiguous, element
 * (q,y,x) at ((q*NY)+y)*NX+x (reference global.h:37-41,112 with COALESCING=true, main.cpp:37-38),
 * unless lbm_set_layout selects the reference's other layout.
 * Where the reference passes scalars as rank-0 device Views (global.h:121-147) this ABI takes plain
 * values; they become kernel arguments.
 *
 * Conventions
 *   - `prec` is sizeof(FLT): LBM_F32 (4) or LBM_F64 (8) -- the reference's USE_SINGLE_PRECISION
 *     switch (global.h:21-27) made a run-time argument.
 *   - `f` is read-only and `buf` write-only within a sweep; they must not overlap.  The caller swaps
 *     them between steps (reference main.cpp:82-84).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All calls are
 *     asynchronous on that stream unless the comment says the call synchronises.
 *   - Pointers are device pointers on the current HIP device, 16-byte aligned.
 *   - Every function returns an lbm_status; nothing throws.  lbm_last_error() gives the detail of the
 *     calling thread's last failure.
 *   - lbm_set_layout, lbm_set_unroll_loops / lbm_set_arithmetic and lbm_set_launch_shape are
 *     process-wide settings (the reference's compile-time macros made run-time switches).  They are
 *     atomic variables read once per call: set them before sweeps are issued from several threads;
 *     a call in flight keeps the value it read.
 *   - x is always periodic inside a *_rows call (a slab spans the full NX); the reference's
 *     2-D halo forms (inner/outer) take explicit bounds and never wrap.
 */
#ifndef LBM_HIP_H
#define LBM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum lbm_status {
    LBM_OK = 0,
    LBM_ERR_INVALID_ARGUMENT = 1, /* null/unaligned pointer, zero extent, bad precision, bad range */
    LBM_ERR_HIP = 2,              /* a HIP runtime call failed; see lbm_last_error() */
    LBM_ERR_NO_DEVICE = 3,        /* no usable gfx950 device */
    LBM_ERR_UNSUPPORTED = 4
} lbm_status;

enum { LBM_F32 = 4, LBM_F64 = 8 };

/* Number of discrete velocities (reference global.h:86). */
enum { LBM_Q = 9 };

const char *lbm_status_string(int status);
const char *lbm_last_error(void);
/* ABI version; bumped when a signature changes. */
int lbm_abi_version(void);

/* The reference's COALESCING switch (macros.h:7, global.h:37-41) as a process-wide run-time setting.
 * LBM_LAYOUT_SOA (default): [9][NY][NX].  LBM_LAYOUT_AOS: [NY][NX][9], element (q,y,x) at
 * ((y*NX)+x)*9+q (main.cpp:34-35), for every distribution array passed to any entry point from
 * then on.  In AOS the initialisers, the whole-lattice sweeps, the pull *_rows forms, every *_fused
 * entry point and the velocity / max-|u| outputs work as documented; the push *_rows forms on a
 * partial row range, the x-halo'd inner/outer forms and rows of one node return
 * LBM_ERR_UNSUPPORTED.  Results are bit-identical to the SOA ones, node by node. */
enum { LBM_LAYOUT_SOA = 0, LBM_LAYOUT_AOS = 1 };
int lbm_set_layout(int layout);
int lbm_get_layout(void);

/* The reference's UNROLL_LOOPS switch (macros.h:6) as a process-wide run-time setting.  It selects
 * between two spellings of the BGK relaxation in the periodic sweeps: pull_periodic / push_periodic
 * (pull-push.cpp:5-73, :124-186; unrolled != 0, the default) and pull_periodic_for /
 * push_periodic_for (pull-push.cpp:77-120, :190-243; unrolled == 0), which round the equilibrium
 * differently (4.5*(c*c) instead of (4.5*c)*c) and so differ in the last bits.  The lid-driven sweeps
 * have one spelling in the reference and ignore the switch.  unrolled == 0 selects LBM_ARITH_ROLLED;
 * unrolled != 0 leaves any other arithmetic as it is and turns LBM_ARITH_ROLLED back into the
 * arithmetic the process started with.  With unrolled == 0 the *_fused entry points need fp32 rows
 * of an even length >= 4, and [NY][NX][9] arrays are not supported. */
int lbm_set_unroll_loops(int unrolled);
int lbm_get_unroll_loops(void);

/* Spelling of the BGK relaxation, process-wide (a superset of lbm_set_unroll_loops):
 *   LBM_ARITH_CONTRACTED  the default.  pull_periodic / push_periodic / push_lid_driven
 *                         (pull-push.cpp:25-72, :138-185, :454-483) with the multiply-adds fused that
 *                         a compiler may fuse under the reference's own build flags (-O3
 *                         -march=native, CMakeLists.txt:18; nvcc and hipcc contract by default):
 *                         ux*ux+uy*uy, 1 -+ 3a, + (4.5a)a, w*(...) - f and f + om*(...) each become
 *                         one fused multiply-add; sums, the reciprocal of the density (correctly
 *                         rounded) and all other products are unchanged.  The contraction is spelled
 *                         out with explicit fma calls, so it does not depend on compiler flags and a
 *                         CPU restatement can follow it bit for bit.  A third fewer vector
 *                         instructions per node than the strict spelling.  Populations stay within
 *                         1e-5 (fp32) / 1e-12 (fp64) max-abs of the strict spelling over thousands of
 *                         steps (measured 3e-7 / 1e-15).  All sweeps in both layouts; fp32 rows must
 *                         have an even length >= 4 for the *_fused entry points (on [NY][NX][9]
 *                         arrays for every sweep).
 *   LBM_ARITH_UNROLLED    the same functions with every operation rounded where the source text puts
 *                         it ("strict"): what a build of the reference with -ffp-contract=off computes.
 *   LBM_ARITH_ROLLED      the *_for functions (pull-push.cpp:77-120, :190-243) = lbm_set_unroll_loops(0).
 * The environment variable LBM_ARITHMETIC (contracted | strict = unrolled | rolled) names the value a
 * process starts with. */
enum { LBM_ARITH_UNROLLED = 0, LBM_ARITH_ROLLED = 1, LBM_ARITH_CONTRACTED = 2 };
int lbm_set_arithmetic(int arithmetic);
int lbm_get_arithmetic(void);

/* Optional launch-shape hint, the counterpart of the reference's -tx/-ty MDRange tile
 * (global.h:95-98, pull-push.cpp:249): threads per workgroup along x (64, 128, 256 or 512) and the
 * cap on resident waves per compute unit for the one-step sweep kernels (1..32; the sweeps stream
 * fastest with about 8-12).  The multi-step sweeps take threads_x as their workgroup width too
 * (threads_x / 64 wavefronts with adjacent strips per workgroup, at most four, where their rings
 * leave room for a group).  0 keeps the built-in choice for that parameter.  Process-wide. */
int lbm_set_launch_shape(unsigned threads_x, unsigned waves_per_cu);

/* Settings that travel with a caller instead of with the process.  The reference fixes layout and
 * arithmetic per build (macros.h:5-8) and its launch tile per run (global.h:95-98); the process-wide
 * setters above mirror that.  A context carries its own copy of the four settings -- layout,
 * arithmetic, threads_x, waves_per_cu -- so that two simulations with different settings can live in
 * one process and a setter called on one thread cannot change what another thread computes.
 *   lbm_ctx_create        new context; its settings start as those in force for the calling thread
 *                         (its current context's, else the process-wide ones).
 *   lbm_ctx_set_* / get_* change / read the context's own settings (same values and errors as the
 *                         process-wide setters).
 *   lbm_ctx_make_current  makes `ctx` the current context of the CALLING HOST THREAD: every entry
 *                         point of this header called from that thread then runs under the
 *                         context's settings; NULL returns the thread to the process-wide settings.
 *                         A thread that never calls it behaves as before.
 *   lbm_ctx_get_current   the calling thread's current context, or NULL.
 *   lbm_ctx_destroy       frees the context (and un-currents it on the calling thread if it is
 *                         current there; other threads must have dropped it before).
 * A context is plain host state: it owns no device memory and no stream, and it may be current on
 * several threads as long as nobody changes its settings meanwhile. */
typedef struct lbm_ctx lbm_ctx;
int lbm_ctx_create(lbm_ctx **out);
int lbm_ctx_destroy(lbm_ctx *ctx);
int lbm_ctx_make_current(lbm_ctx *ctx);
lbm_ctx *lbm_ctx_get_current(void);
int lbm_ctx_set_layout(lbm_ctx *ctx, int layout);
int lbm_ctx_set_arithmetic(lbm_ctx *ctx, int arithmetic);
int lbm_ctx_set_launch_shape(lbm_ctx *ctx, unsigned threads_x, unsigned waves_per_cu);
int lbm_ctx_get_layout(const lbm_ctx *ctx);      /* -1 for a null context */
int lbm_ctx_get_arithmetic(const lbm_ctx *ctx);  /* -1 for a null context */

/* Tuning and test switches of the multi-step launches (process-wide; no counterpart in the
 * reference).  The environment variables LBM_FUSED_NARROW, LBM_FUSED_GROUP, LBM_FUSED_BULK,
 * LBM_FUSED_MAX_STEPS and LBM_FUSED_MIN_NODES name their start values, read once when the library is loaded; this call
 * changes one afterwards.  name: "fused_narrow" (0 never / 1 where it saves a strip / 2 always:
 * halos of whole lanes on small periodic lattices), "fused_group" (-1 built-in / 0 single
 * wavefronts / 1 groups of wavefronts wherever the kernel allows), "fused_bulk" (0: every
 * wavefront of a cavity launch runs the code with the wall cases), "fused_max_steps" (0 built-in,
 * else the most steps lbm_plan_steps hands to one launch), "fused_min_nodes" (0 none, else
 * lattices with fewer nodes take one step per launch). */
int lbm_set_tuning(const char *name, int value);

/* ---------------------------------------------------------------- initialisation ----------- */

/* Replaces init_shearwave (init.h:13, init.cpp:5-34).  Rows [halo, ny-halo) of f get the
 * equilibrium of ux = u*sinf(2*pi*(y-halo+y_below)/y_glob), uy = 0; halo is the reference's
 * USE_MPI offset (0 or 1 there, init.cpp:7-12; any ghost depth with 2*halo <= ny here).  The per-row equilibrium table is evaluated on the host
 * with libm sinf and double arithmetic exactly as init.cpp:57-84 spells it, then broadcast along x
 * on the device.  Synchronises the stream. */
int lbm_init_shearwave(void *f, int prec, uint64_t nx, uint64_t ny, double rho_init, double u,
                       uint64_t y_below, uint64_t y_glob, unsigned halo, void *stream);

/* Replaces init_rest (init.h:18, init.cpp:36-49): every row of f gets w_q*rho_init. */
int lbm_init_rest(void *f, int prec, uint64_t nx, uint64_t ny, double rho_init, void *stream);

/* ---------------------------------------------------------------- periodic sweeps ---------- */

/* Replaces pull_periodic_no_mpi (pull-push.h:84, pull-push.cpp:268-288). */
int lbm_pull_periodic_no_mpi(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny,
                             double om, void *stream);
/* Replaces push_periodic_no_mpi (pull-push.h:73, pull-push.cpp:246-266). */
int lbm_push_periodic_no_mpi(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny,
                             double om, void *stream);

/* Slab forms: rows [y_lo,y_hi) of an [9][ny][nx] array, x periodic.  wrap_y != 0 wraps the y
 * neighbours inside the array (pull-push.cpp:277-280); wrap_y == 0 reads/writes rows y-1 and y+1
 * as they are (ghost rows present: requires y_lo >= 1 and y_hi <= ny-1), which is the y part of
 * pull_periodic_inner / push_periodic_inner (pull-push.h:95,107). */
int lbm_pull_periodic_rows(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                           uint64_t y_lo, uint64_t y_hi, int wrap_y, void *stream);
int lbm_push_periodic_rows(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                           uint64_t y_lo, uint64_t y_hi, int wrap_y, void *stream);

/* Replace pull_periodic_inner / push_periodic_inner (pull-push.h:95,107; pull-push.cpp:291-332):
 * rectangle [LX,HX) x [LY,HY) of a halo'd array, no wrap (requires 1<=LX, HX<=nx-1, 1<=LY,
 * HY<=ny-1). */
int lbm_pull_periodic_inner(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                            uint64_t LX, uint64_t LY, uint64_t HX, uint64_t HY, void *stream);
int lbm_push_periodic_inner(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                            uint64_t LX, uint64_t LY, uint64_t HX, uint64_t HY, void *stream);
/* Replace pull_periodic_outer / push_periodic_outer (pull-push.h:117,126; pull-push.cpp:335-429):
 * the ring x=1, x=nx-2, y=1, y=ny-2 of a halo'd array (requires nx,ny >= 4). */
int lbm_pull_periodic_outer(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                            void *stream);
int lbm_push_periodic_outer(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                            void *stream);

/* ---------------------------------------------------------------- lid-driven cavity -------- */

/* Replaces push_lid_driven (pull-push.h:140, pull-push.cpp:432-531): push sweep with half-way
 * bounce-back on all four walls and the moving lid above y=ny-1 (df = rho_eq*u_lid/6). */
int lbm_push_lid_driven(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                        double rho_eq, double u_lid, void *stream);
/* Pull form of the same boundary conditions.  No counterpart in the reference (its lid scene always
 * pushes, main.cpp:72-74); the stored state is post-collision as in pull_periodic. */
int lbm_pull_lid_driven(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                        double rho_eq, double u_lid, void *stream);
/* Slab forms: rows [y_lo,y_hi) of the array; array row r is global row r + y_goff of a cavity
 * ny_glob rows high, so the bottom wall / lid act only where the global row is 0 / ny_glob-1. */
int lbm_push_lid_rows(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                      double rho_eq, double u_lid, uint64_t y_lo, uint64_t y_hi, int64_t y_goff,
                      uint64_t ny_glob, void *stream);
int lbm_pull_lid_rows(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                      double rho_eq, double u_lid, uint64_t y_lo, uint64_t y_hi, int64_t y_goff,
                      uint64_t ny_glob, void *stream);

/* ------------------------------------------------ several time steps per call ------------ */

/* Each of the following replaces `nsteps` consecutive iterations of the reference's time loop
 * (main.cpp:64-83 single rank: sweep, swap, sweep, ...; main.cpp:217-396 with halos) for rows
 * [y_lo, y_hi): it reads f at time t and leaves time t + nsteps in buf, bit-identical to nsteps calls
 * of the one-step entry point with the buffers swapped in between, while every population crosses
 * HBM once.  nsteps = 1 .. lbm_fused_max_steps().  Rows outside [y_lo, y_hi) of buf are not written
 * (the one-step push entry points scatter into the neighbouring rows instead).  nsteps = 1 is
 * forwarded to the one-step entry point, with its semantics, when the range is the whole lattice
 * or the row length is not supported here.
 *
 * Rows that the dependency cone of the updated rows reaches must be present in the array: with
 * wrap_y = 0 the periodic forms read rows [y_lo - nsteps, y_hi + nsteps) (deep ghost rows: copies of
 * the neighbouring slab's rows at time t), the lid forms the same range clipped at the walls.
 * Returns LBM_ERR_UNSUPPORTED (nothing launched) for rows of a single node; callers then take
 * nsteps one-step sweeps. */

/* Makes the runtime load, on the current device, the code objects of the sweeps that the settings
 * in force select (layout, arithmetic) for this precision.  HIP otherwise loads a code object at the
 * first launch of one of its kernels, a few milliseconds that then fall on that launch.  The
 * reference pays the same inside Kokkos::initialize (main.cpp:412), ahead of its clock (:421).
 * Optional; launches nothing. */
int lbm_load_kernels(int prec);

/* Largest nsteps the fused entry points accept for this precision and row length: 8, or 1 when the
 * row length is unsupported. */
int lbm_fused_max_steps(int prec, uint64_t nx);

/* How many of the `remaining` iterations of the time loop the next call should take for a lattice
 * of `rows` rows of nx nodes: the run is cut into the fewest launches of at most
 * lbm_fused_max_steps() steps (fewer on small lattices; 1 for row lengths the fused sweeps do not
 * support) and those launches are made as equal as possible (20 remaining -> 7, then 7, 6).  Environment: LBM_FUSED_MAX_STEPS caps the answer,
 * LBM_FUSED_MIN_NODES makes lattices below that many nodes take one step per launch.  The time loops of this library's drivers (main.cpp:64-83) ask this before
 * every launch. */
int lbm_plan_steps(int prec, uint64_t nx, uint64_t rows, uint64_t remaining);
/* The same for a named scene: lid != 0 plans the lid-driven cavity sweeps (lbm_*_lid_fused).  The two
 * scenes are planned alike at present (the cavity kernels run the periodic sweep's code in every
 * wavefront that is eight nodes or more from a wall); callers that know their scene should say so.
 * lbm_plan_steps() is this with lid = 0. */
int lbm_plan_steps_scene(int prec, uint64_t nx, uint64_t rows, uint64_t remaining, int lid);

/* nsteps x pull_periodic (pull-push.cpp:5-73, :268-288). */
int lbm_pull_periodic_fused(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                            uint64_t y_lo, uint64_t y_hi, int wrap_y, int nsteps, void *stream);
/* nsteps x push_periodic (pull-push.cpp:124-186, :246-266). */
int lbm_push_periodic_fused(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                            uint64_t y_lo, uint64_t y_hi, int wrap_y, int nsteps, void *stream);
/* nsteps x push_lid_driven (pull-push.cpp:436-528, :290-309); row offset and cavity height as in
 * lbm_push_lid_rows. */
int lbm_push_lid_fused(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                       double rho_eq, double u_lid, uint64_t y_lo, uint64_t y_hi, int64_t y_goff,
                       uint64_t ny_glob, int nsteps, void *stream);
/* nsteps x the pull form of the lid-driven cavity (lbm_pull_lid_rows). */
int lbm_pull_lid_fused(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                       double rho_eq, double u_lid, uint64_t y_lo, uint64_t y_hi, int64_t y_goff,
                       uint64_t ny_glob, int nsteps, void *stream);

/* ---------------------------------------------------------------- output path -------------- */

/* Replaces compute_velocities (io.cpp:136-174).  vel is a device array [2][y_hi-y_lo][nx] (ux plane,
 * then uy plane), so stores are coalesced; the reference's vel(x,y,c) indexing carries no
 * arithmetic. */
int lbm_compute_velocities(const void *f, void *vel, int prec, uint64_t nx, uint64_t ny,
                           uint64_t y_lo, uint64_t y_hi, void *stream);
/* |u| per node as the -ov output forms it (io.cpp:236-238): sqrtf of the FLT sum of squares.
 * mag is a device array [y_hi-y_lo][nx]. */
int lbm_velocity_magnitudes(const void *f, void *mag, int prec, uint64_t nx, uint64_t ny,
                            uint64_t y_lo, uint64_t y_hi, void *stream);
/* Replaces the MAX_VEL reduction of output() (io.cpp:184-208): max over rows [y_lo,y_hi) of
 * sqrtf(ux^2+uy^2), NaNs ignored, -1e30 when no node qualifies.  The result is written as one FLT
 * to out_dev (device memory, 8-byte aligned); asynchronous.  The result word doubles as the
 * reduction's scratch (it holds intermediate values until the call's last kernel has run), so the
 * call allocates nothing and calls with different out_dev never interfere. */
int lbm_max_velocity_dev(const void *f, void *out_dev, int prec, uint64_t nx, uint64_t ny,
                         uint64_t y_lo, uint64_t y_hi, void *stream);
/* No counterpart in the reference.  lbm_max_velocity_dev for callers that cannot allocate device
 * memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): same
 * arguments, same kernels, same result, but the reduction's key lives in a 256-byte array that is part
 * of the library's code object (one copy per device, held until the process ends) instead of a lazily
 * hipMalloc'ed word, and the call makes no hipMalloc / hipFree.  The scratch need does not grow with
 * the lattice, so the size limits are the ABI's own (extents up to 2^31-1 per axis, rejected with
 * LBM_ERR_INVALID_ARGUMENT beyond).  The first call on a device looks the array's address up
 * (hipGetSymbolAddress): make that call outside a graph capture.  Every call on a device shares that
 * device's array, which is cleared on `stream` at the start of each call: calls that overlap on one
 * device (different streams or threads) race. */
int lbm_max_velocity_dev_static(const void *f, void *out_dev, int prec, uint64_t nx, uint64_t ny,
                                uint64_t y_lo, uint64_t y_hi, void *stream);
/* Same as lbm_max_velocity_dev, result delivered to the host as double (exact widening of the FLT
 * value).  Synchronises the stream.  Its device scratch is per host thread and device (allocated on
 * the thread's first call), so concurrent calls from several threads do not interfere. */
int lbm_max_velocity(const void *f, int prec, uint64_t nx, uint64_t ny, uint64_t y_lo,
                     uint64_t y_hi, void *stream, double *out_host);

/* ---------------------------------------------------------------- self-check --------------- */

/* No counterpart in the reference.  out[i] = the reciprocal of x[i] exactly as LBM_ARITH_CONTRACTED
 * forms 1/rho (pull-push.cpp:27): in fp32 a hardware reciprocal estimate and one fused Newton step,
 * which must equal the correctly rounded IEEE quotient 1.0f / x[i] for every normal x; in fp64 the
 * IEEE division itself.  x and out are device arrays of n FLT.  The test-suite runs it over every
 * fp32 value of three binades. */
int lbm_selftest_reciprocal(const void *x, void *out, int prec, uint64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LBM_HIP_H */
