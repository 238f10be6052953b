// C ABI (include/lbm_hip.h) over the gfx950 kernels of lbm_device.h.
//
// Host-side duties only: argument validation (a faulting kernel can take the whole node down, so
// every shape the kernels assume is checked here), choice between the 128-bit vector kernels and
// the scalar any-size kernels, launch geometry, the host-evaluated equilibrium tables of the
// initialisers, and error reporting.  No computation on populations happens on the host.
#include "lbm_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>
#include <vector>

#include "lbm_device.h"
#include "lbm_fused.h"
#include "lbm_fused_instances.h"

#pragma clang fp contract(off)

// the multi-step kernels are compiled in lbm_fused_inst_*.hip, side by side with this file
LBM_FUSED_ALL_SETS(LBM_FUSED_DECLARE)

// Settings that travel with a caller instead of with the process (include/lbm_hip.h, lbm_ctx_*).
struct lbm_ctx {
    int layout;
    int arith;
    unsigned threads_x, waves_per_cu;
};

namespace {

thread_local char g_err[512] = "";

int fail(int status, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return status;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(LBM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                                  \
    } while (0)

// memory layout of every distribution array passed in (lbm_set_layout)
std::atomic<int> g_layout{LBM_LAYOUT_SOA};

// spelling of the relaxation (lbm_set_arithmetic / lbm_set_unroll_loops): lbm::kArithContracted (the
// default: the reference is built with contraction allowed, CMakeLists.txt:18), kArithUnrolled
// (LBM_ARITHMETIC=strict: every operation rounded where pull-push.cpp:25-72 puts it) or kArithRolled
// (the reference's UNROLL_LOOPS=false spelling; periodic sweeps only).  The environment variable
// names the default a process starts with.
int arithmetic_default()
{
    const char *e = getenv("LBM_ARITHMETIC");
    if (e && (strcmp(e, "strict") == 0 || strcmp(e, "unrolled") == 0)) return lbm::kArithUnrolled;
    if (e && strcmp(e, "rolled") == 0) return lbm::kArithRolled;
    return lbm::kArithContracted;
}
std::atomic<int> g_arith{arithmetic_default()};

// launch-shape hint (lbm_set_launch_shape)
std::atomic<unsigned> g_threads_x{0};
std::atomic<unsigned> g_waves_per_cu{0};

// The settings a call runs under: those of the context that is current on the calling host thread
// (lbm_ctx_make_current), else the process-wide ones above.  Every entry point reads them through
// these accessors, once per decision.
thread_local lbm_ctx *t_ctx = nullptr;
int cur_layout() { return t_ctx ? t_ctx->layout : g_layout.load(); }
int cur_arith() { return t_ctx ? t_ctx->arith : g_arith.load(); }
unsigned cur_threads_x() { return t_ctx ? t_ctx->threads_x : g_threads_x.load(); }
unsigned cur_waves_per_cu() { return t_ctx ? t_ctx->waves_per_cu : g_waves_per_cu.load(); }

// Measured on MI355X (16384^2, straight-line sweeps, nt stores, 64-lane workgroups): the achieved
// HBM rate is flat at 4-6 resident waves per CU (fp32 pull 6.06-6.13 TB/s), falls off a cliff at 7
// (5.6 TB/s; 8 and more 5.4-5.6) and collapses at 3 (3.8 TB/s, latency exposed).  A wave keeps 9 KiB
// of loads and 9 KiB of stores in flight, so 7 waves x 32 CUs exceed the 4 MiB L2 of an XCD.
// Residency is capped by giving every workgroup an unused dynamic-LDS allocation of
// 160 KiB / (workgroups per CU).
constexpr unsigned kDefaultThreadsX = 64;
constexpr unsigned kWavesPerCU_f32 = 6;       // top of the plateau for fp32 pull and push (4096^2 .. 32768x8192)
constexpr unsigned kWavesPerCU_f64_pull = 5;  // fp64 pull: 4 is best by 1 % but sits on the cliff to 3
constexpr unsigned kWavesPerCU_f64_push = 6;
constexpr unsigned kWavesPerCU_cache_resident = 16;  // fp64 push (aligned-store form): 6 beats 5 by 4 %
constexpr size_t kLdsPerCU = 160 * 1024;   // gfx950, in every partition mode (partitions have fewer units, not smaller ones)
constexpr size_t kMaxDynamicLds = 64 * 1024;

// Re-deal workgroups so that one XCD walks whole rows (lbm::block_coords).  Off by default:
// measured on MI355X (16384^2) it removes the 8 % FETCH_SIZE excess of the plain mapping but lowers
// the achieved rate (fp32 pull 6.1 -> 5.7 TB/s, fp64 pull 5.8 -> 5.6); the excess lines are cheap
// (served on-die), the lost address-order locality is not.  A finer variant (one row at a time, each
// XCD a contiguous eighth of the row) lost even more (6.16 -> 5.38 TB/s).  LBM_XCD_ROWS=1 enables it.
int xcd_rows_default()
{
    const char *e = getenv("LBM_XCD_ROWS");
    return e ? atoi(e) : 0;
}
int g_xcd_rows = xcd_rows_default();

constexpr int kMaxDevicesForProps = 64;
// lattices below this many nodes (1.5 M) are bound by the chain of dependent iterations of a
// wavefront, not by memory: five steps per launch (lbm_plan_steps), narrow halos (launch_fused)
constexpr uint64_t kSmallLatticeNodes = 3ull << 19;
constexpr uint64_t kMaxExtent = 0x7fffffffull;  // extents are 32-bit inside the kernels

int check_common(const void *f, int prec, uint64_t nx, uint64_t ny)
{
    if (prec != LBM_F32 && prec != LBM_F64)
        return fail(LBM_ERR_INVALID_ARGUMENT, "precision must be 4 (fp32) or 8 (fp64), got %d", prec);
    if (f == nullptr) return fail(LBM_ERR_INVALID_ARGUMENT, "null population pointer");
    if ((uintptr_t)f % 16 != 0)
        return fail(LBM_ERR_INVALID_ARGUMENT, "population pointer %p is not 16-byte aligned", f);
    if (nx == 0 || ny == 0) return fail(LBM_ERR_INVALID_ARGUMENT, "empty lattice %llux%llu",
                                        (unsigned long long)nx, (unsigned long long)ny);
    if (nx > kMaxExtent || ny > kMaxExtent)
        return fail(LBM_ERR_INVALID_ARGUMENT, "lattice extent beyond 2^31-1: %llux%llu",
                    (unsigned long long)nx, (unsigned long long)ny);
    return LBM_OK;
}

int check_pair(const void *f, const void *buf, int prec, uint64_t nx, uint64_t ny)
{
    if (int s = check_common(f, prec, nx, ny)) return s;
    if (buf == nullptr) return fail(LBM_ERR_INVALID_ARGUMENT, "null write-buffer pointer");
    if ((uintptr_t)buf % 16 != 0)
        return fail(LBM_ERR_INVALID_ARGUMENT, "write-buffer pointer %p is not 16-byte aligned", buf);
    const uint64_t bytes = 9ull * nx * ny * (uint64_t)prec;
    const uintptr_t a = (uintptr_t)f, b = (uintptr_t)buf;
    if ((a <= b && b < a + bytes) || (b <= a && a < b + bytes))
        return fail(LBM_ERR_INVALID_ARGUMENT, "f and buf overlap");
    return LBM_OK;
}

int check_rows(uint64_t ny, uint64_t y_lo, uint64_t y_hi, bool needs_ghosts)
{
    if (y_lo > y_hi || y_hi > ny)
        return fail(LBM_ERR_INVALID_ARGUMENT, "row range [%llu,%llu) outside [0,%llu)",
                    (unsigned long long)y_lo, (unsigned long long)y_hi, (unsigned long long)ny);
    if (needs_ghosts && y_lo < y_hi && (y_lo < 1 || y_hi + 1 > ny))
        return fail(LBM_ERR_INVALID_ARGUMENT,
                    "row range [%llu,%llu) needs ghost rows y_lo-1 and y_hi inside [0,%llu)",
                    (unsigned long long)y_lo, (unsigned long long)y_hi, (unsigned long long)ny);
    return LBM_OK;
}

unsigned pick_block(uint64_t lanes_per_row)
{
    const unsigned tx = cur_threads_x();
    unsigned b = tx ? tx : kDefaultThreadsX;
    // do not launch workgroups much wider than a row
    while (b > 64 && (uint64_t)b / 2 >= lanes_per_row) b /= 2;
    return b;
}

// Grid of the sweep kernels: one workgroup per (x chunk, row); rows are split over grid y and z
// (lbm::block_row).
dim3 sweep_grid(uint64_t lanes_per_row, unsigned block, uint64_t rows)
{
    const uint64_t gy = rows < lbm::kRowsPerGridZ ? rows : lbm::kRowsPerGridZ;
    const uint64_t gz = (rows + lbm::kRowsPerGridZ - 1) / lbm::kRowsPerGridZ;
    return dim3((unsigned)((lanes_per_row + block - 1) / block), (unsigned)gy, (unsigned)gz);
}

// Dynamic-LDS bytes per workgroup that leave room for `waves` resident waves per CU.
size_t residency_ballast(unsigned block, size_t elem_size, bool pull, uint64_t nx, uint64_t ny)
{
    // both arrays resident in the 256 MiB Infinity Cache: the L2 cliff does not bite and more waves
    // hide more latency (1024^2 fp32: 66.3 GLUPS at 16-24 waves/CU against 64.5 at 6)
    const bool cache_resident = 2ull * 9ull * nx * ny * elem_size <= (192ull << 20);
    const unsigned wpc = cur_waves_per_cu();
    const unsigned waves = wpc ? wpc
                           : cache_resident ? kWavesPerCU_cache_resident
                           : (elem_size == 4 ? kWavesPerCU_f32 : (pull ? kWavesPerCU_f64_pull : kWavesPerCU_f64_push));
    const unsigned waves_per_block = block / 64;
    unsigned blocks_per_cu = waves / waves_per_block;
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    if (waves >= 32) return 0;  // hardware limit anyway
    size_t bytes = (kLdsPerCU / blocks_per_cu) & ~(size_t)511;
    if (bytes > kMaxDynamicLds) bytes = kMaxDynamicLds;
    return bytes;
}

// Grid of the row-strided helper kernels (initialisation, velocity output).
dim3 make_grid(uint64_t lanes_per_row, unsigned block, uint64_t rows)
{
    uint64_t gy = rows;
    if (gy > 65535) gy = 65535;
    if (gy == 0) gy = 1;
    return dim3((unsigned)((lanes_per_row + block - 1) / block), (unsigned)gy, 1);
}

lbm::Geom make_geom(uint64_t nx, uint64_t ny, uint64_t y_lo, uint64_t y_hi, int wrap_y,
                    int64_t y_goff = 0, uint64_t ny_glob = 0)
{
    lbm::Geom g;
    g.xcd_rows = g_xcd_rows;
    g.nx = (uint32_t)nx;
    g.ny = (uint32_t)ny;
    g.y_lo = (uint32_t)y_lo;
    g.y_hi = (uint32_t)y_hi;
    g.wrap_y = wrap_y;
    g.y_goff = y_goff;
    g.ny_glob = ny_glob;
    return g;
}

int launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LBM_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return LBM_OK;
}

// Dispatch a kernel template over the supported workgroup widths.
#define LBM_FOR_BLOCK(block, ...)                         \
    switch (block) {                                      \
        case 64:  { constexpr int B = 64;  __VA_ARGS__; } break;  \
        case 128: { constexpr int B = 128; __VA_ARGS__; } break;  \
        case 512: { constexpr int B = 512; __VA_ARGS__; } break;  \
        default:  { constexpr int B = 256; __VA_ARGS__; } break;  \
    }

template <typename T> constexpr int vec_width() { return 16 / (int)sizeof(T); }

// One step on [NY][NX][9] arrays; defined with the multi-step sweeps, whose kernels it runs.
template <bool PULL, bool LID>
int aos_one_step(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om, double rho_eq,
                 double u_lid, uint64_t y_lo, uint64_t y_hi, int wrap_y, int64_t y_goff, uint64_t ny_glob,
                 void *stream);
const char *const kAosNoHalo =
    "the x-halo'd inner/outer forms exist only for [9][NY][NX] arrays (the reference's MPI path too, main.cpp:147)";

// ---- periodic sweeps -------------------------------------------------------------------
template <typename T, bool PULL, int ARITH>
int sweep_periodic_rows_as(const void *f_, void *buf_, uint64_t nx, uint64_t ny, double om_,
                        uint64_t y_lo, uint64_t y_hi, int wrap_y, hipStream_t stream)
{
    if (y_lo == y_hi) return LBM_OK;
    const T *f = (const T *)f_;
    T *buf = (T *)buf_;
    const T om = (T)om_;  // reference io.cpp:112
    constexpr int V = vec_width<T>();
    const lbm::Geom g = make_geom(nx, ny, y_lo, y_hi, wrap_y);
    const uint64_t rows = y_hi - y_lo;
    if (nx % V == 0 && nx >= 2 * V) {
        const uint64_t lanes = nx / V;
        const unsigned block = pick_block(lanes);
        const dim3 grid = sweep_grid(lanes, block, rows);
        const size_t lds = residency_ballast(block, sizeof(T), PULL, nx, ny);
        LBM_FOR_BLOCK(block,
            if (PULL) hipLaunchKernelGGL((lbm::k_pull_periodic_vec<T, V, B, ARITH>), grid, dim3(B), lds, stream, f, buf, g, om);
            else      hipLaunchKernelGGL((lbm::k_push_periodic_vec<T, V, B, ARITH>), grid, dim3(B), lds, stream, f, buf, g, om))
    } else if (nx >= 4 * V) {
        // rows that are not a whole number of 16-byte vectors: element-aligned vector kernel
        const uint64_t lanes = (nx + V - 1) / V;
        const unsigned block = pick_block(lanes);
        const dim3 grid = sweep_grid(lanes, block, rows);
        const size_t lds = residency_ballast(block, sizeof(T), PULL, nx, ny);
        constexpr int MODE = PULL ? lbm::kPullPeriodic : lbm::kPushPeriodic;
        LBM_FOR_BLOCK(block,
            hipLaunchKernelGGL((lbm::k_sweep_vec_any<T, V, MODE, B, ARITH>), grid, dim3(B), lds, stream, f, buf, g, om, (T)0))
    } else {
        const unsigned block = pick_block(nx);
        const dim3 grid = sweep_grid(nx, block, rows);
        constexpr int MODE = PULL ? lbm::kPullPeriodic : lbm::kPushPeriodic;
        LBM_FOR_BLOCK(block,
            hipLaunchKernelGGL((lbm::k_sweep_scalar<T, MODE, B, ARITH>), grid, dim3(B), 0, stream, f, buf, g,
                               0u, (uint32_t)nx, 1, om, (T)0))
    }
    return launch_status(PULL ? "pull periodic" : "push periodic");
}

// every spelling of the relaxation runs the same kernels (vector, element-aligned vector, one node per lane)
template <typename T, bool PULL>
int sweep_periodic_rows(const void *f, void *buf, uint64_t nx, uint64_t ny, double om, uint64_t y_lo, uint64_t y_hi,
                        int wrap_y, hipStream_t stream)
{
    switch (cur_arith()) {
        case lbm::kArithRolled: return sweep_periodic_rows_as<T, PULL, lbm::kArithRolled>(f, buf, nx, ny, om, y_lo, y_hi, wrap_y, stream);
        case lbm::kArithContracted: return sweep_periodic_rows_as<T, PULL, lbm::kArithContracted>(f, buf, nx, ny, om, y_lo, y_hi, wrap_y, stream);
        default: return sweep_periodic_rows_as<T, PULL, lbm::kArithUnrolled>(f, buf, nx, ny, om, y_lo, y_hi, wrap_y, stream);
    }
}

template <bool PULL>
int periodic_rows(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                  uint64_t y_lo, uint64_t y_hi, int wrap_y, void *stream)
{
    if (int s = check_pair(f, buf, prec, nx, ny)) return s;
    if (int s = check_rows(ny, y_lo, y_hi, !wrap_y)) return s;
    if (cur_layout() == LBM_LAYOUT_AOS) return aos_one_step<PULL, false>(f, buf, prec, nx, ny, om, 0., 0., y_lo, y_hi, wrap_y, 0, 0, stream);
    hipStream_t st = (hipStream_t)stream;
    return prec == LBM_F32
               ? sweep_periodic_rows<float, PULL>(f, buf, nx, ny, om, y_lo, y_hi, wrap_y, st)
               : sweep_periodic_rows<double, PULL>(f, buf, nx, ny, om, y_lo, y_hi, wrap_y, st);
}

// ---- halo'd 2-D forms --------------------------------------------------------------------
template <typename T, bool PULL, int ARITH>
int sweep_inner_as(const void *f_, void *buf_, uint64_t nx, uint64_t ny, double om_, uint64_t LX,
                uint64_t LY, uint64_t HX, uint64_t HY, hipStream_t stream)
{
    if (LX >= HX || LY >= HY) return LBM_OK;
    const lbm::Geom g = make_geom(nx, ny, LY, HY, 0);
    constexpr int V = vec_width<T>();
    if (PULL && HX - LX >= 4 * (uint64_t)V) {
        const uint64_t lanes = (HX - LX + V - 1) / V;
        const unsigned block = pick_block(lanes);
        const dim3 grid = sweep_grid(lanes, block, HY - LY);
        const size_t lds = residency_ballast(block, sizeof(T), true, nx, ny);
        LBM_FOR_BLOCK(block,
            hipLaunchKernelGGL((lbm::k_pull_inner_vec<T, V, B, ARITH>), grid, dim3(B), lds, stream, (const T *)f_,
                               (T *)buf_, g, (uint32_t)LX, (uint32_t)HX, (T)om_))
        return launch_status("pull periodic inner");
    }
    const unsigned block = pick_block(HX - LX);
    const dim3 grid = sweep_grid(HX - LX, block, HY - LY);
    constexpr int MODE = PULL ? lbm::kPullPeriodic : lbm::kPushPeriodic;
    LBM_FOR_BLOCK(block,
        hipLaunchKernelGGL((lbm::k_sweep_scalar<T, MODE, B, ARITH>), grid, dim3(B), 0, stream, (const T *)f_,
                           (T *)buf_, g, (uint32_t)LX, (uint32_t)HX, 0, (T)om_, (T)0))
    return launch_status("periodic inner");
}

template <typename T, bool PULL>
int sweep_inner(const void *f, void *buf, uint64_t nx, uint64_t ny, double om, uint64_t LX, uint64_t LY, uint64_t HX,
                uint64_t HY, hipStream_t stream)
{
    switch (cur_arith()) {
        case lbm::kArithRolled: return sweep_inner_as<T, PULL, lbm::kArithRolled>(f, buf, nx, ny, om, LX, LY, HX, HY, stream);
        case lbm::kArithContracted: return sweep_inner_as<T, PULL, lbm::kArithContracted>(f, buf, nx, ny, om, LX, LY, HX, HY, stream);
        default: return sweep_inner_as<T, PULL, lbm::kArithUnrolled>(f, buf, nx, ny, om, LX, LY, HX, HY, stream);
    }
}

template <bool PULL>
int inner(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om, uint64_t LX,
          uint64_t LY, uint64_t HX, uint64_t HY, void *stream)
{
    if (int s = check_pair(f, buf, prec, nx, ny)) return s;
    if (cur_layout() == LBM_LAYOUT_AOS) return fail(LBM_ERR_UNSUPPORTED, "%s", kAosNoHalo);
    if (HX > nx || HY > ny || LX > HX || LY > HY)
        return fail(LBM_ERR_INVALID_ARGUMENT, "rectangle [%llu,%llu)x[%llu,%llu) outside %llux%llu",
                    (unsigned long long)LX, (unsigned long long)HX, (unsigned long long)LY,
                    (unsigned long long)HY, (unsigned long long)nx, (unsigned long long)ny);
    if (LX < HX && LY < HY && (LX < 1 || HX + 1 > nx || LY < 1 || HY + 1 > ny))
        return fail(LBM_ERR_INVALID_ARGUMENT, "inner rectangle needs a halo of one node on every side");
    hipStream_t st = (hipStream_t)stream;
    return prec == LBM_F32 ? sweep_inner<float, PULL>(f, buf, nx, ny, om, LX, LY, HX, HY, st)
                           : sweep_inner<double, PULL>(f, buf, nx, ny, om, LX, LY, HX, HY, st);
}

template <typename T, bool PULL>
int sweep_outer(const void *f_, void *buf_, uint64_t nx, uint64_t ny, double om_, hipStream_t stream)
{
    const uint64_t lanes = (nx - 2) + (ny - 4);
    constexpr int B = 256;
    if (cur_arith() == lbm::kArithRolled)
        hipLaunchKernelGGL((lbm::k_outer_ring<T, PULL, B, lbm::kArithRolled>), dim3((unsigned)((lanes + B - 1) / B)), dim3(B), 0,
                           stream, (const T *)f_, (T *)buf_, (uint32_t)nx, (uint32_t)ny, (T)om_);
    else if (cur_arith() == lbm::kArithContracted)
        hipLaunchKernelGGL((lbm::k_outer_ring<T, PULL, B, lbm::kArithContracted>), dim3((unsigned)((lanes + B - 1) / B)), dim3(B), 0,
                           stream, (const T *)f_, (T *)buf_, (uint32_t)nx, (uint32_t)ny, (T)om_);
    else
        hipLaunchKernelGGL((lbm::k_outer_ring<T, PULL, B>), dim3((unsigned)((lanes + B - 1) / B)), dim3(B), 0,
                           stream, (const T *)f_, (T *)buf_, (uint32_t)nx, (uint32_t)ny, (T)om_);
    return launch_status("periodic outer");
}

template <bool PULL>
int outer(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om, void *stream)
{
    if (int s = check_pair(f, buf, prec, nx, ny)) return s;
    if (cur_layout() == LBM_LAYOUT_AOS) return fail(LBM_ERR_UNSUPPORTED, "%s", kAosNoHalo);
    if (nx < 4 || ny < 4)
        return fail(LBM_ERR_INVALID_ARGUMENT, "outer ring needs nx,ny >= 4 (got %llux%llu)",
                    (unsigned long long)nx, (unsigned long long)ny);
    hipStream_t st = (hipStream_t)stream;
    return prec == LBM_F32 ? sweep_outer<float, PULL>(f, buf, nx, ny, om, st)
                           : sweep_outer<double, PULL>(f, buf, nx, ny, om, st);
}

// ---- lid-driven cavity ---------------------------------------------------------------------
template <typename T, bool PULL, int ARITH>
int sweep_lid_rows_as(const void *f_, void *buf_, uint64_t nx, uint64_t ny, double om_, double rho_,
                   double u_, uint64_t y_lo, uint64_t y_hi, int64_t y_goff, uint64_t ny_glob,
                   hipStream_t stream)
{
    if (y_lo == y_hi) return LBM_OK;
    const T *f = (const T *)f_;
    T *buf = (T *)buf_;
    const T om = (T)om_;
    // df = RHO_EQ * U_LID / 6 in FLT (pull-push.cpp:516-518)
    const T rho = (T)rho_, u = (T)u_;
    const T df = rho * u / (T)6.;
    constexpr int V = vec_width<T>();
    const lbm::Geom g = make_geom(nx, ny, y_lo, y_hi, 0, y_goff, ny_glob);
    const uint64_t rows = y_hi - y_lo;
    if (nx % V == 0 && nx >= 2 * V) {
        const uint64_t lanes = nx / V;
        const unsigned block = pick_block(lanes);
        const dim3 grid = sweep_grid(lanes, block, rows);
        const size_t lds = residency_ballast(block, sizeof(T), PULL, nx, ny);
        LBM_FOR_BLOCK(block,
            hipLaunchKernelGGL((lbm::k_lid_vec<T, V, B, PULL, ARITH>), grid, dim3(B), lds, stream, f, buf, g, om, df))
    } else if (nx >= 4 * V) {
        const uint64_t lanes = (nx + V - 1) / V;
        const unsigned block = pick_block(lanes);
        const dim3 grid = sweep_grid(lanes, block, rows);
        const size_t lds = residency_ballast(block, sizeof(T), PULL, nx, ny);
        constexpr int MODE = PULL ? lbm::kPullLid : lbm::kPushLid;
        LBM_FOR_BLOCK(block,
            hipLaunchKernelGGL((lbm::k_sweep_vec_any<T, V, MODE, B, ARITH>), grid, dim3(B), lds, stream, f, buf, g, om, df))
    } else {
        const unsigned block = pick_block(nx);
        const dim3 grid = sweep_grid(nx, block, rows);
        constexpr int MODE = PULL ? lbm::kPullLid : lbm::kPushLid;
        LBM_FOR_BLOCK(block,
            hipLaunchKernelGGL((lbm::k_sweep_scalar<T, MODE, B, ARITH>), grid, dim3(B), 0, stream, f, buf, g,
                               0u, (uint32_t)nx, 0, om, df))
    }
    return launch_status(PULL ? "pull lid-driven" : "push lid-driven");
}

// the cavity sweep has one spelling in the reference and its contraction; the rolled switch does not apply
template <typename T, bool PULL>
int sweep_lid_rows(const void *f, void *buf, uint64_t nx, uint64_t ny, double om, double rho, double u, uint64_t y_lo,
                   uint64_t y_hi, int64_t y_goff, uint64_t ny_glob, hipStream_t stream)
{
    if (cur_arith() == lbm::kArithContracted)
        return sweep_lid_rows_as<T, PULL, lbm::kArithContracted>(f, buf, nx, ny, om, rho, u, y_lo, y_hi, y_goff, ny_glob, stream);
    return sweep_lid_rows_as<T, PULL, lbm::kArithUnrolled>(f, buf, nx, ny, om, rho, u, y_lo, y_hi, y_goff, ny_glob, stream);
}

template <bool PULL>
int lid_rows(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om, double rho_eq,
             double u_lid, uint64_t y_lo, uint64_t y_hi, int64_t y_goff, uint64_t ny_glob,
             void *stream)
{
    if (int s = check_pair(f, buf, prec, nx, ny)) return s;
    if (int s = check_rows(ny, y_lo, y_hi, false)) return s;
    if (y_lo < y_hi) {
        // every updated row must lie inside the cavity, and rows that are not wall rows need their
        // y neighbours inside the array
        const int64_t g_lo = (int64_t)y_lo + y_goff, g_hi = (int64_t)y_hi + y_goff;
        if (ny_glob == 0 || ny_glob > kMaxExtent || g_lo < 0 || (uint64_t)g_hi > ny_glob)
            return fail(LBM_ERR_INVALID_ARGUMENT,
                        "rows [%llu,%llu) + offset %lld leave the cavity of height %llu",
                        (unsigned long long)y_lo, (unsigned long long)y_hi, (long long)y_goff,
                        (unsigned long long)ny_glob);
        if ((g_lo != 0 && y_lo == 0) || ((uint64_t)g_hi != ny_glob && y_hi == ny))
            return fail(LBM_ERR_INVALID_ARGUMENT,
                        "slab rows that are not wall rows need a ghost row inside the array");
    }
    if (cur_layout() == LBM_LAYOUT_AOS)
        return aos_one_step<PULL, true>(f, buf, prec, nx, ny, om, rho_eq, u_lid, y_lo, y_hi, 0, y_goff, ny_glob, stream);
    hipStream_t st = (hipStream_t)stream;
    return prec == LBM_F32
               ? sweep_lid_rows<float, PULL>(f, buf, nx, ny, om, rho_eq, u_lid, y_lo, y_hi, y_goff, ny_glob, st)
               : sweep_lid_rows<double, PULL>(f, buf, nx, ny, om, rho_eq, u_lid, y_lo, y_hi, y_goff, ny_glob, st);
}

// ---- several time steps per pass (lbm_fused.h) ------------------------------------------------
// Rows stored per wavefront.  Measured on MI355X: long segments amortise the K-1 (pull) or K (push)
// rows of redundant work at either end, short ones keep more wavefronts in flight; at 16384^2 fp32
// (16-byte lanes) the best were 16 rows for K=2 (147 GLUPS), 32 for K=3 (193), 64 for K=4 (217); with
// 8-byte lanes K=4 is flat from 64 to 128 rows (+1 % at 128) and loses 4 % at 256.  Small grids choose
// by a model of rounds x iterations (launch_fused).
unsigned fused_seg_rows_default()
{
    const char *e = getenv("LBM_FUSED_SEG_ROWS");
    return e ? (unsigned)atoi(e) : 0u;
}
unsigned g_fused_seg_rows = fused_seg_rows_default();
int fused_xcd_default()
{
    const char *e = getenv("LBM_FUSED_XCD");
    return e ? atoi(e) : 0;
}
int g_fused_xcd = fused_xcd_default();
// LBM_FUSED_WALL_SPLIT=0: wall strips of small cavities keep the segment length of the other strips
int fused_wall_split_default()
{
    const char *e = getenv("LBM_FUSED_WALL_SPLIT");
    return e ? atoi(e) : 1;
}
int g_fused_wall_split = fused_wall_split_default();

// Tuning and test switches of the multi-step launches: the environment names their start values
// (read once, when the library is loaded), lbm_set_tuning changes them afterwards.
int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
std::atomic<int> g_fused_narrow{env_int("LBM_FUSED_NARROW", 1)};
std::atomic<int> g_fused_group{env_int("LBM_FUSED_GROUP", -1)};
std::atomic<int> g_fused_bulk{env_int("LBM_FUSED_BULK", 1)};
std::atomic<int> g_fused_max_steps{env_int("LBM_FUSED_MAX_STEPS", 0)};
std::atomic<int> g_fused_min_nodes{env_int("LBM_FUSED_MIN_NODES", 0)};
std::atomic<int> g_fused_lane_nodes{env_int("LBM_FUSED_LANE_NODES", 0)};

int max_fused_steps(int /*prec*/) { return lbm::kFusedMaxSteps; }

// Wavefronts the device holds at once at two per SIMD (the residency of every multi-step kernel):
// compute units x 4 SIMDs x 2.  2048 on a whole MI355X; a partition (CPX, DPX) has fewer units.
uint64_t wave_slots()
{
    static std::atomic<uint64_t> slots[kMaxDevicesForProps];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevicesForProps) return 2048;
    uint64_t n = slots[dev].load();
    if (n == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        n = (uint64_t)cus * 8;
        slots[dev].store(n);
    }
    return n;
}

template <typename T> constexpr int fused_width() { return lbm::kFusedLaneBytes / (int)sizeof(T); }

// Rows of odd length in fp32 cannot be cut into 8-byte lanes; they run with one node per lane
// (48 stored lanes per strip).  Anything from two nodes per row up is supported.
template <typename T>
bool fused_supported(uint64_t nx)
{
    return nx >= 2;
}

template <typename T, int V, bool PUSH, bool LID, bool AOS, int ARITH = lbm::kArithUnrolled>
int launch_fused(const T *f, T *buf, uint64_t nx, uint64_t ny, T om, T df, uint64_t y_lo, uint64_t y_hi,
                 int wrap_y, int64_t y_goff, uint64_t ny_glob, int k, hipStream_t stream)
{
    // lanes stored per strip: 56, or 48 for fp64 beyond four steps (wider x halo)
    static_assert(lbm::fused_halo_lanes<T, V, 5>() == lbm::fused_halo_lanes<T, V, lbm::kFusedMaxSteps>(),
                  "one halo width for every step count above four");
    // rows are addressed through buffer descriptors with 32-bit byte offsets
    if (nx * sizeof(T) >= (uint64_t)lbm::kNoStore)
        return fail(LBM_ERR_UNSUPPORTED, "rows of %llu nodes are too long for the multi-step sweeps", (unsigned long long)nx);
    const uint64_t rows = y_hi - y_lo;
    // Periodic lattices small enough to live in the caches take their launches of five steps (what
    // lbm_plan_steps gives them) with halos of whole lanes instead of whole 32-byte granules where
    // that makes fewer strips (lbm::fused_halo_lanes): 1024^2 fp32 144 -> 153 GLUPS, fp64 88 -> 91;
    // with the same number of strips the ragged line ends cost 2-4 % (512^2, 768^2), and the cavity
    // gains nothing (1024^2 push 121 -> 119).  LBM_FUSED_NARROW=0 (or lbm_set_tuning) keeps the
    // granule, 2 takes narrow halos whatever the strip count.
    const uint64_t W_granule = lbm::kWave - 2 * (k <= 4 ? lbm::fused_halo_lanes<T, V, 4>()
                                                        : lbm::fused_halo_lanes<T, V, lbm::kFusedMaxSteps>());
    const uint64_t W_narrow = lbm::kWave - 2 * lbm::fused_halo_lanes<T, V, lbm::kFusedNarrowSteps, true>();
    const int narrow_mode = g_fused_narrow.load();
    const bool narrow = k == lbm::kFusedNarrowSteps && !AOS && !LID && V * sizeof(T) == (size_t)lbm::kFusedLaneBytes &&
                        nx * rows < kSmallLatticeNodes && narrow_mode != 0 &&
                        (narrow_mode == 2 || (nx / V + W_narrow - 1) / W_narrow < (nx / V + W_granule - 1) / W_granule);
    const uint64_t W = narrow ? W_narrow : W_granule;
    const uint64_t nstrips = (nx / V + W - 1) / W;
    uint64_t seg = g_fused_seg_rows ? g_fused_seg_rows : (k <= 2 ? 16 : k == 3 ? 32 : 128);
    bool small_grid = false;
    if (!g_fused_seg_rows) {
        // A wavefront walks seg + (rows of y halo) iterations one after the other, and the chip holds
        // `slots` wavefronts at a time.  Grids of many rounds use the measured segment length; smaller
        // ones pick the length that minimises rounds x iterations -- the tail round of a 2.4-round grid
        // is as expensive as a full one (fp32 pull, GLUPS before -> after: 512^2 38 -> 46, 2048^2
        // 137 -> 174, 4096^2 180 -> 224, 8192^2 231 -> 253; 1024^2 stays at 107, launch-gap-bound).
        // Slots: two wavefronts per SIMD.  A few kernels fit three per SIMD (fp32 push, three to
        // five steps); counting those made their grids worse (1024^2 push 100 -> 80 GLUPS): a third
        // wavefront shares the SIMD's issue slots, it does not add a slot's worth of speed.
        const uint64_t slots = wave_slots();
        const uint64_t extra = (PUSH || AOS) ? 2 * (uint64_t)k : 2 * (uint64_t)(k - 1);
        if (nstrips * ((rows + seg - 1) / seg) < 8 * slots) {
            small_grid = true;
            // A wavefront alone on its SIMD finishes an iteration 1.6 x sooner than one that shares it
            // (1024^2 fp32, six steps: 1.75 us against 3.0 us per iteration), so a last round that
            // leaves every SIMD at most one wavefront is priced at 1.0 per iteration and a fuller one
            // at 1.6: 1024^2 takes 12-row segments (860 wavefronts, 22 iterations) rather than 6-row
            // ones (1710 wavefronts, 16 iterations), 119 -> 146 GLUPS.
            const uint64_t simds = slots / 2;
            uint64_t best = ~0ull;
            for (uint64_t s = 186; s >= 2; --s) {
                const uint64_t waves = nstrips * ((rows + s - 1) / s);
                const uint64_t full = waves / slots, rest = waves % slots;
                const uint64_t cost = (full * 16 + (rest == 0 ? 0 : rest <= simds ? 10 : 16)) * (s + extra);
                if (cost < best) { best = cost; seg = s; }
            }
        } else if (seg == 128) {
            // Larger grids: the same count, over the lengths at which a wavefront is as efficient
            // as at 128 rows.  16384^2 in 128-row segments is 9.2 rounds, in 118-row ones 9.98:
            // fp32 pull 296 -> 305 GLUPS, push 294 -> 301, cavity push 300 -> 309.
            // Seven and eight steps have more rows of y halo to amortise: 140 to 224 rows there
            // (16384^2 fp32, eight steps, GLUPS at 100/118/140/160/186/223/279/372 rows:
            // 406/418/420/423/424/423/417/404).  The wider the range, the better the rounds of a slab
            // come out: 4080 rows of 32768 nodes are 3.72 rounds at 158 rows, 2.86 at 204.
            uint64_t best = ~0ull;
            for (uint64_t s = (k >= 7 ? 224 : 128); s >= (k >= 7 ? 140u : 96u); --s) {
                const uint64_t waves = nstrips * ((rows + s - 1) / s);
                const uint64_t cost = ((waves + slots - 1) / slots) * (s + extra);
                if (cost < best) { best = cost; seg = s; }
            }
        }
    }
    if (seg > rows) seg = rows;
    const uint64_t nsegs = (rows + seg - 1) / seg;
    if (nstrips * nsegs > 0x7fffffffull)
        return fail(LBM_ERR_INVALID_ARGUMENT, "lattice too large for one fused launch");
    // Cavities on grids of a round or two: the wavefronts of the two wall strips run the code with
    // the wall cases in every iteration, about a fifth slower than the others (1024^2 fp32, four
    // steps: 44 against 36 us per launch), and a small grid is over when its slowest wavefront is.
    // They get segments short enough to finish with everyone else.
    // LBM_FUSED_GROUP (tests and tuning): 0 = one wavefront per workgroup everywhere, 1 = groups
    // wherever the kernel's rings allow them, whatever the lattice size
    const int group_mode = g_fused_group.load();
    uint64_t seg_wall = 0, nsegs_wall = 0;
    if (LID && small_grid && !g_fused_xcd && g_fused_wall_split && group_mode <= 0) {
        const uint64_t extra = (PUSH || AOS) ? 2 * (uint64_t)k : 2 * (uint64_t)(k - 1);
        const uint64_t iters = (seg + extra) * 5 / 6;          // iterations at 1.2 x the time each
        const uint64_t slots = wave_slots(), nwall = nstrips < 2 ? nstrips : 2;
        const uint64_t rounds = (nstrips * nsegs + slots - 1) / slots;
        // the shortest segments that do not push the grid into another round of wavefronts -- nor,
        // where every wavefront has a SIMD to itself, beyond one wavefront per SIMD
        const uint64_t cap = nstrips * nsegs <= slots / 2 ? slots / 2 : rounds * slots;
        for (uint64_t s = iters > extra + 2 ? iters - extra : 2; s < seg; ++s) {
            const uint64_t nw = (rows + s - 1) / s;
            if ((nstrips - nwall) * nsegs + nwall * nw <= cap) {
                seg_wall = s;
                nsegs_wall = nw;
                break;
            }
        }
    }
    lbm::FusedGeom g;
    g.nx = (uint32_t)nx; g.ny = (uint32_t)ny; g.y_lo = (uint32_t)y_lo; g.y_hi = (uint32_t)y_hi;
    g.wrap_y = wrap_y; g.seg_rows = (uint32_t)seg; g.nstrips = (uint32_t)nstrips; g.nsegs = (uint32_t)nsegs;
    g.y_goff = y_goff; g.ny_glob = ny_glob;
    g.xcd_map = g_fused_xcd;
    // LBM_FUSED_BULK=0 (tests): every wavefront of a cavity launch runs the code with the wall cases
    g.bulk_code = g_fused_bulk.load();
    g.seg_rows_wall = (uint32_t)seg_wall; g.nsegs_wall = (uint32_t)nsegs_wall;
    const uint64_t nsegs_grid = g.xcd_map ? (nsegs + 7) / 8 * 8 : nsegs;
    const uint64_t nwall = nstrips < 2 ? nstrips : 2;
    const uint64_t blocks = nsegs_wall ? (nstrips - nwall) * nsegs + nwall * nsegs_wall : nstrips * nsegs_grid;
    // Wavefronts per workgroup (lbm::fused_group_waves): groups on lattices of many rounds, where
    // neighbouring strips that start together share their common lines in the cache; single
    // wavefronts, free to go to any compute unit, on small ones.  A group's LDS can exceed the
    // 64 KiB a kernel may ask for without saying so beforehand.
    const bool grouped = nsegs_wall == 0 && !g.xcd_map && (group_mode < 0 ? !small_grid : group_mode > 0);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevicesForProps) dev = 0;
    const unsigned hint_tx = cur_threads_x();
    const unsigned hint_waves = hint_tx == 0 ? 0u : (hint_tx / 64 > (unsigned)lbm::kFusedMaxGroupWaves ? (unsigned)lbm::kFusedMaxGroupWaves : hint_tx / 64);
#define LBM_FUSED_CASE(KK)                                                                                       \
    case KK: {                                                                                                   \
        unsigned nw = grouped ? (unsigned)lbm::fused_group_waves<T, V, KK, PUSH, LID, AOS, ARITH>() : 1u;        \
        /* the launch-shape hint (the reference's -tx tile width): workgroups of threads_x / 64 wavefronts where */ \
        /* the kernel's rings leave room for a group at all                                                      */ \
        if (hint_waves != 0 && nsegs_wall == 0 && !g.xcd_map &&                                                   \
            lbm::fused_group_waves<T, V, KK, PUSH, LID, AOS, ARITH>() > 1)                                        \
            nw = hint_waves;                                                                                      \
        const unsigned lds = nw * (unsigned)lbm::fused_lds_bytes_per_wave<T, V, KK, PUSH, LID, AOS, ARITH>();    \
        auto kern = lbm::k_fused<T, V, KK, PUSH, LID, AOS, ARITH>;                                               \
        if constexpr (KK == lbm::kFusedNarrowSteps && !AOS && !LID && V * sizeof(T) == (size_t)lbm::kFusedLaneBytes) { \
            if (narrow) kern = lbm::k_fused<T, V, KK, PUSH, LID, AOS, ARITH, true>;                              \
        }                                                                                                        \
        /* one flag per kernel function and device: the narrow and the granule kernel are two functions */      \
        static std::atomic<bool> lds_allowed_of[2][kMaxDevicesForProps];                                         \
        std::atomic<bool> *lds_allowed = lds_allowed_of[narrow ? 1 : 0];                                         \
        if (lds > 64 * 1024 && !lds_allowed[dev].load()) {                                                       \
            const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                     (int)lds);                                                  \
            if (e != hipSuccess)                                                                                 \
                return fail(LBM_ERR_HIP, "hipFuncSetAttribute(%u bytes of LDS): %s", lds, hipGetErrorString(e)); \
            lds_allowed[dev].store(true);                                                                        \
        }                                                                                                        \
        hipLaunchKernelGGL(kern, dim3((unsigned)((blocks + nw - 1) / nw)), dim3(lbm::kWave * nw), lds, stream,   \
                           f, buf, g, om, df);                                                                   \
        break;                                                                                                   \
    }
    switch (k) {
        LBM_FUSED_CASE(1)
        LBM_FUSED_CASE(2)
        LBM_FUSED_CASE(3)
        LBM_FUSED_CASE(4)
        LBM_FUSED_CASE(5)
        LBM_FUSED_CASE(6)
        LBM_FUSED_CASE(7)
        LBM_FUSED_CASE(8)
        default: return fail(LBM_ERR_INVALID_ARGUMENT, "fused step count %d outside 1..%d", k, max_fused_steps(0));
    }
#undef LBM_FUSED_CASE
    return launch_status("fused sweep");
}

template <typename T, bool PUSH, bool LID>
int sweep_fused(const void *f_, void *buf_, uint64_t nx, uint64_t ny, double om_, double rho_, double u_,
                uint64_t y_lo, uint64_t y_hi, int wrap_y, int64_t y_goff, uint64_t ny_glob, int k,
                hipStream_t stream)
{
    if (y_lo == y_hi) return LBM_OK;
    const T *f = (const T *)f_;
    T *buf = (T *)buf_;
    const T om = (T)om_;
    const T rho = (T)rho_, u = (T)u_;
    const T df = LID ? rho * u / (T)6. : (T)0.;   // pull-push.cpp:516-518
    constexpr int V8 = fused_width<T>();   // eight bytes of a row per lane
    // one node per lane: fp32 rows that cannot be cut into 8-byte lanes, or the tuning switch fused_lane_nodes = 1
    const bool narrow = V8 > 1 && (nx % V8 != 0 || nx < 2 * V8 ||
                                   (g_fused_lane_nodes.load() == 1 && cur_arith() == lbm::kArithUnrolled));
    if constexpr (!LID) {
        if (cur_arith() == lbm::kArithRolled) {
            if (cur_layout() == LBM_LAYOUT_AOS)
                return fail(LBM_ERR_UNSUPPORTED, "the rolled arithmetic is not available on [NY][NX][9] arrays");
            if (narrow)
                return fail(LBM_ERR_UNSUPPORTED,
                            "the rolled arithmetic takes several steps per call only for fp32 rows of an even length >= 4 (nx=%llu)",
                            (unsigned long long)nx);
            return launch_fused<T, V8, PUSH, false, false, lbm::kArithRolled>(f, buf, nx, ny, om, df, y_lo, y_hi, wrap_y, y_goff, ny_glob, k, stream);
        }
    }
    if (cur_arith() == lbm::kArithContracted) {
        if (narrow)
            return fail(LBM_ERR_UNSUPPORTED,
                        "the contracted arithmetic takes several steps per call only for fp32 rows of an even length >= 4 (nx=%llu)",
                        (unsigned long long)nx);
        if (cur_layout() == LBM_LAYOUT_AOS)
            return launch_fused<T, V8, PUSH, LID, true, lbm::kArithContracted>(f, buf, nx, ny, om, df, y_lo, y_hi, wrap_y, y_goff, ny_glob, k, stream);
        return launch_fused<T, V8, PUSH, LID, false, lbm::kArithContracted>(f, buf, nx, ny, om, df, y_lo, y_hi, wrap_y, y_goff, ny_glob, k, stream);
    }
    if (cur_layout() == LBM_LAYOUT_AOS) {
        if constexpr (V8 > 1) {
            if (narrow)
                return launch_fused<T, 1, PUSH, LID, true>(f, buf, nx, ny, om, df, y_lo, y_hi, wrap_y, y_goff, ny_glob, k, stream);
        }
        return launch_fused<T, V8, PUSH, LID, true>(f, buf, nx, ny, om, df, y_lo, y_hi, wrap_y, y_goff, ny_glob, k, stream);
    }
    if constexpr (V8 > 1) {
        if (narrow)
            return launch_fused<T, 1, PUSH, LID, false>(f, buf, nx, ny, om, df, y_lo, y_hi, wrap_y, y_goff, ny_glob, k, stream);
    }
    return launch_fused<T, V8, PUSH, LID, false>(f, buf, nx, ny, om, df, y_lo, y_hi, wrap_y, y_goff, ny_glob, k, stream);
}

template <bool PULL>
int periodic_fused(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                   uint64_t y_lo, uint64_t y_hi, int wrap_y, int k, void *stream)
{
    if (k < 1 || k > max_fused_steps(prec))
        return fail(LBM_ERR_INVALID_ARGUMENT, "step count %d outside 1..%d", k, max_fused_steps(prec));
    if (int s = check_pair(f, buf, prec, nx, ny)) return s;
    const bool supported = lbm_fused_max_steps(prec, nx) > 1;
    // one step of the whole periodic lattice is the one-step sweep's home ground; so is any row
    // length the fused kernels cannot take
    if (k == 1 && cur_layout() == LBM_LAYOUT_SOA &&
        (!supported || (wrap_y && y_lo == 0 && y_hi == ny)))
        return periodic_rows<PULL>(f, buf, prec, nx, ny, om, y_lo, y_hi, wrap_y, stream);
    if (int s = check_rows(ny, y_lo, y_hi, false)) return s;
    if (!wrap_y && y_lo < y_hi && (y_lo < (uint64_t)k || y_hi + (uint64_t)k > ny))
        return fail(LBM_ERR_INVALID_ARGUMENT,
                    "%d fused steps of rows [%llu,%llu) need %d ghost rows on either side inside [0,%llu)",
                    k, (unsigned long long)y_lo, (unsigned long long)y_hi, k, (unsigned long long)ny);
    if (!supported)
        return fail(LBM_ERR_UNSUPPORTED, "fused sweeps need rows of at least two nodes%s (nx=%llu)",
                    cur_arith() != lbm::kArithUnrolled ? ", and an even number of fp32 nodes with the rolled or contracted arithmetic" : "",
                    (unsigned long long)nx);
    hipStream_t st = (hipStream_t)stream;
    return prec == LBM_F32
               ? sweep_fused<float, !PULL, false>(f, buf, nx, ny, om, 0., 0., y_lo, y_hi, wrap_y, 0, 0, k, st)
               : sweep_fused<double, !PULL, false>(f, buf, nx, ny, om, 0., 0., y_lo, y_hi, wrap_y, 0, 0, k, st);
}

template <bool PULL>
int lid_fused(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om, double rho_eq,
              double u_lid, uint64_t y_lo, uint64_t y_hi, int64_t y_goff, uint64_t ny_glob, int k,
              void *stream)
{
    if (k < 1 || k > max_fused_steps(prec))
        return fail(LBM_ERR_INVALID_ARGUMENT, "step count %d outside 1..%d", k, max_fused_steps(prec));
    if (int s = check_pair(f, buf, prec, nx, ny)) return s;
    {
        const bool supported = prec == LBM_F32 ? fused_supported<float>(nx) : fused_supported<double>(nx);
        const bool whole = y_lo == 0 && y_hi == ny && y_goff == 0 && ny_glob == ny;
        if (k == 1 && cur_layout() == LBM_LAYOUT_SOA && (!supported || whole))
            return lid_rows<PULL>(f, buf, prec, nx, ny, om, rho_eq, u_lid, y_lo, y_hi, y_goff, ny_glob, stream);
    }
    if (int s = check_rows(ny, y_lo, y_hi, false)) return s;
    if (y_lo < y_hi) {
        const int64_t g_lo = (int64_t)y_lo + y_goff, g_hi = (int64_t)y_hi + y_goff;
        if (ny_glob == 0 || ny_glob > kMaxExtent || g_lo < 0 || (uint64_t)g_hi > ny_glob)
            return fail(LBM_ERR_INVALID_ARGUMENT,
                        "rows [%llu,%llu) + offset %lld leave the cavity of height %llu",
                        (unsigned long long)y_lo, (unsigned long long)y_hi, (long long)y_goff,
                        (unsigned long long)ny_glob);
        // rows of the cavity within k rows of the range must be present in the array
        const uint64_t need_lo = (uint64_t)g_lo < (uint64_t)k ? (uint64_t)g_lo : (uint64_t)k;
        const uint64_t above = ny_glob - (uint64_t)g_hi;
        const uint64_t need_hi = above < (uint64_t)k ? above : (uint64_t)k;
        if (y_lo < need_lo || ny - y_hi < need_hi)
            return fail(LBM_ERR_INVALID_ARGUMENT,
                        "%d fused steps of rows [%llu,%llu) need %llu ghost rows below and %llu above inside the array",
                        k, (unsigned long long)y_lo, (unsigned long long)y_hi,
                        (unsigned long long)need_lo, (unsigned long long)need_hi);
    }
    if (!(prec == LBM_F32 ? fused_supported<float>(nx) : fused_supported<double>(nx)))
        return fail(LBM_ERR_UNSUPPORTED, "fused sweeps need rows of at least two nodes (nx=%llu)",
                    (unsigned long long)nx);
    hipStream_t st = (hipStream_t)stream;
    return prec == LBM_F32
               ? sweep_fused<float, !PULL, true>(f, buf, nx, ny, om, rho_eq, u_lid, y_lo, y_hi, 0, y_goff, ny_glob, k, st)
               : sweep_fused<double, !PULL, true>(f, buf, nx, ny, om, rho_eq, u_lid, y_lo, y_hi, 0, y_goff, ny_glob, k, st);
}

// One-step entry points on [NY][NX][9] arrays run the multi-step kernels with one step.  Those
// gather into the rows of the range; the one-step push sweeps scatter out of it, which is the same
// thing only for a whole lattice.
template <bool PULL, bool LID>
int aos_one_step(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om, double rho_eq,
                 double u_lid, uint64_t y_lo, uint64_t y_hi, int wrap_y, int64_t y_goff, uint64_t ny_glob,
                 void *stream)
{
    if (nx < 2)
        return fail(LBM_ERR_UNSUPPORTED, "[NY][NX][9] arrays need rows of at least two nodes (nx=%llu)",
                    (unsigned long long)nx);
    if (!LID && cur_arith() == lbm::kArithRolled)
        return fail(LBM_ERR_UNSUPPORTED, "the rolled arithmetic is not available on [NY][NX][9] arrays");
    if (cur_arith() == lbm::kArithContracted && prec == LBM_F32 && (nx % 2 != 0 || nx < 4))
        return fail(LBM_ERR_UNSUPPORTED,
                    "the contracted arithmetic on [NY][NX][9] arrays needs fp32 rows of an even length >= 4 (nx=%llu)",
                    (unsigned long long)nx);
    const bool whole = y_lo == 0 && y_hi == ny && (LID ? (y_goff == 0 && ny_glob == ny) : wrap_y != 0);
    if (!PULL && !whole)
        return fail(LBM_ERR_UNSUPPORTED,
                    "push sweeps of a row range of [NY][NX][9] arrays: use the *_fused entry points with nsteps = 1");
    hipStream_t st = (hipStream_t)stream;
    return prec == LBM_F32
               ? sweep_fused<float, !PULL, LID>(f, buf, nx, ny, om, rho_eq, u_lid, y_lo, y_hi, wrap_y, y_goff, ny_glob, 1, st)
               : sweep_fused<double, !PULL, LID>(f, buf, nx, ny, om, rho_eq, u_lid, y_lo, y_hi, wrap_y, y_goff, ny_glob, 1, st);
}

// ---- initialisation --------------------------------------------------------------------
// f_eq of init.cpp:57-84, evaluated on the host.  The bracket is double even for T=float because
// the reference writes 9./2. and 3./2. (init.cpp:81-82); 1+3*cu, cu*cu and u_2 are formed in T.
template <typename T>
T f_eq_host(T ux, T uy, T rho_i, unsigned dir)
{
    static const T w[9] = {(T)(4. / 9.),  (T)(1. / 9.),  (T)(1. / 9.),  (T)(1. / 9.), (T)(1. / 9.),
                           (T)(1. / 36.), (T)(1. / 36.), (T)(1. / 36.), (T)(1. / 36.)};
    static const T cx[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};
    static const T cy[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
    const T cx_ux = cx[dir] * ux;
    const T cy_uy = cy[dir] * uy;
    const T c_i_u_r = cx_ux + cy_uy;
    const T u_2 = ux * ux + uy * uy;
    const T a = T(1) + T(3) * c_i_u_r;
    const T cc = c_i_u_r * c_i_u_r;
    const double bracket = ((double)a + 9. / 2. * (double)cc) - 3. / 2. * (double)u_2;
    const T wr = w[dir] * rho_i;
    return (T)((double)wr * bracket);
}

template <typename T>
int fill_from_table(void *f_, uint64_t nx, uint64_t ny, uint64_t y_first, const std::vector<T> &table,
                    uint64_t rows, hipStream_t stream)
{
    if (rows == 0) return LBM_OK;
    T *d_table = nullptr;
    HIP_TRY(hipMalloc((void **)&d_table, table.size() * sizeof(T)));
    hipError_t e = hipMemcpyAsync(d_table, table.data(), table.size() * sizeof(T), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) {
        const unsigned block = pick_block(nx);
        const dim3 grid = make_grid(nx, block, rows);
        if (cur_layout() == LBM_LAYOUT_AOS) {
            LBM_FOR_BLOCK(block,
                hipLaunchKernelGGL((lbm::k_fill_rows<T, B, true>), grid, dim3(B), 0, stream, (T *)f_, (uint32_t)nx,
                                   (uint32_t)ny, (uint32_t)y_first, (uint32_t)rows, (const T *)d_table))
        } else if (nx % vec_width<T>() == 0) {
            constexpr int V = vec_width<T>();
            const dim3 vgrid = make_grid(nx / V, 64, rows);
            hipLaunchKernelGGL((lbm::k_fill_rows_vec<T, V, 64>), vgrid, dim3(64), 0, stream, (T *)f_, (uint32_t)nx,
                               (uint32_t)ny, (uint32_t)y_first, (uint32_t)rows, (const T *)d_table);
        } else {
            LBM_FOR_BLOCK(block,
                hipLaunchKernelGGL((lbm::k_fill_rows<T, B>), grid, dim3(B), 0, stream, (T *)f_, (uint32_t)nx,
                                   (uint32_t)ny, (uint32_t)y_first, (uint32_t)rows, (const T *)d_table))
        }
        e = hipGetLastError();
    }
    // the table (host vector and device copy) must outlive the kernel
    hipError_t e2 = hipStreamSynchronize(stream);
    hipError_t e3 = hipFree(d_table);
    if (e != hipSuccess) return fail(LBM_ERR_HIP, "initialisation failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(LBM_ERR_HIP, "initialisation failed: %s", hipGetErrorString(e2));
    if (e3 != hipSuccess) return fail(LBM_ERR_HIP, "hipFree failed: %s", hipGetErrorString(e3));
    return LBM_OK;
}

template <typename T>
int init_shearwave_t(void *f, uint64_t nx, uint64_t ny, double rho_, double u_, uint64_t y_below,
                     uint64_t y_glob, unsigned halo, hipStream_t stream)
{
    const uint64_t rows = ny - 2ull * halo;
    const T rho_i = (T)rho_, u_init = (T)u_;  // io.cpp:113-114
    std::vector<T> table(9 * rows);
    for (uint64_t r = 0; r < rows; ++r) {
        // init.cpp:22-26 with y = r + halo: (y - use_mpi + y_below) in unsigned arithmetic
        const uint64_t y = r + halo;
        const T y_scaled = ((T)(y - (uint64_t)halo + y_below)) / ((T)y_glob);
        const T ux = u_init * sinf((float)(2.0 * M_PI * y_scaled));
        for (unsigned dir = 0; dir < 9; ++dir) table[dir * rows + r] = f_eq_host<T>(ux, (T)0., rho_i, dir);
    }
    return fill_from_table<T>(f, nx, ny, halo, table, rows, stream);
}

template <typename T>
int init_rest_t(void *f, uint64_t nx, uint64_t ny, double rho_, hipStream_t stream)
{
    const T rho_i = (T)rho_;
    std::vector<T> table(9 * ny);
    for (unsigned dir = 0; dir < 9; ++dir) {
        const T v = f_eq_host<T>((T)0., (T)0., rho_i, dir);
        for (uint64_t r = 0; r < ny; ++r) table[dir * ny + r] = v;
    }
    return fill_from_table<T>(f, nx, ny, 0, table, ny, stream);
}

// ---- output path -----------------------------------------------------------------------
template <typename T, int WHAT>
int velocity_t(const void *f, void *out, uint64_t nx, uint64_t ny, uint64_t y_lo, uint64_t y_hi,
               hipStream_t stream)
{
    if (y_lo == y_hi) return LBM_OK;
    const unsigned block = pick_block(nx);
    const dim3 grid = make_grid(nx, block, y_hi - y_lo);
    if (cur_layout() == LBM_LAYOUT_AOS) {
        LBM_FOR_BLOCK(block,
            hipLaunchKernelGGL((lbm::k_velocity<T, WHAT, B, true>), grid, dim3(B), 0, stream, (const T *)f, (T *)out,
                               (uint32_t)nx, (uint32_t)ny, (uint32_t)y_lo, (uint32_t)y_hi))
    } else {
        LBM_FOR_BLOCK(block,
            hipLaunchKernelGGL((lbm::k_velocity<T, WHAT, B>), grid, dim3(B), 0, stream, (const T *)f, (T *)out,
                               (uint32_t)nx, (uint32_t)ny, (uint32_t)y_lo, (uint32_t)y_hi))
    }
    return launch_status("compute velocities");
}

// Scratch of lbm_max_velocity (key word + decoded result, one 256-byte line): per device and per
// host thread, allocated on a thread's first call on a device.  The call synchronises its stream
// before it returns, so one thread never has two reductions in flight on a line, and two threads
// never share one.  (lbm_max_velocity_dev needs no scratch: the result word itself is the key.)
constexpr int kMaxDevices = 64;
// freed when the host thread ends (a pool of short-lived threads must not leave a line per thread behind)
struct KeyLines {
    uint32_t *line[kMaxDevices] = {};
    ~KeyLines()
    {
        for (uint32_t *p : line)
            if (p != nullptr) (void)hipFree(p);   // any device may be current: hipFree finds the owner
    }
};
thread_local KeyLines t_key;

int key_scratch(uint32_t **out)
{
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(LBM_ERR_NO_DEVICE, "device ordinal %d out of range", dev);
    if (t_key.line[dev] == nullptr) HIP_TRY(hipMalloc((void **)&t_key.line[dev], 256));
    *out = t_key.line[dev];
    return LBM_OK;
}

// The reduction's scratch for callers that cannot allocate (lbm_max_velocity_dev_static): one
// 256-byte line per device, part of the code object, so every device the library runs on holds its
// own copy at its own address until the process ends.  `g_static_scratch` is only a host-side handle
// of the symbol; device code and HIP calls get the per-device address resolved by static_scratch().
constexpr size_t kStaticScratchBytes = 256;
static __device__ __align__(256) uint32_t g_static_scratch[kStaticScratchBytes / sizeof(uint32_t)];
uint32_t *g_static_scratch_ptr[kMaxDevices] = {};
std::mutex g_static_scratch_mutex;

int static_scratch(uint32_t **out)
{
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(LBM_ERR_NO_DEVICE, "device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_static_scratch_mutex);
    if (g_static_scratch_ptr[dev] == nullptr) {
        void *p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_scratch)));
        if (p == nullptr || (uintptr_t)p % 256 != 0)
            return fail(LBM_ERR_HIP, "static scratch of device %d resolved to %p", dev, p);
        g_static_scratch_ptr[dev] = (uint32_t *)p;
    }
    *out = g_static_scratch_ptr[dev];
    return LBM_OK;
}

// Rows of workgroups of the reduction's grid (each walks rows gy apart).  16384^2 fp32, 1024 / 2048 /
// 4096 / 8192 / 16384: 5.8 / 6.0 / 6.15 / 6.2 / 6.2 TB/s (fp64 5.95 ... 6.1) now that a workgroup
// issues its atomic only if it can raise the key; with one atomic per workgroup 4096 was the best
// (6.05 TB/s) and 16384 ran at 2.95.
constexpr uint64_t kMaxSpeedGridRows = 8192;

// `key`: one zeroable word of device scratch (the first of `key_bytes` bytes that are cleared here)
template <typename T>
int max_velocity_t(const void *f, void *out_dev, uint64_t nx, uint64_t ny, uint64_t y_lo,
                   uint64_t y_hi, uint32_t *key, size_t key_bytes, hipStream_t stream)
{
    HIP_TRY(hipMemsetAsync(key, 0, key_bytes, stream));
    if (y_lo < y_hi) {
        constexpr int B = 256;
        constexpr int V = vec_width<T>();
        const uint64_t rows = y_hi - y_lo;
        if (cur_layout() == LBM_LAYOUT_AOS) {
            uint64_t gx = (nx + B - 1) / B;
            if (gx > 64) gx = 64;
            uint64_t gy = rows < kMaxSpeedGridRows ? rows : kMaxSpeedGridRows;
            hipLaunchKernelGGL((lbm::k_max_speed<T, 1, B, true>), dim3((unsigned)gx, (unsigned)gy), dim3(B), 0, stream,
                               (const T *)f, (uint32_t)nx, (uint32_t)ny, (uint32_t)y_lo, (uint32_t)y_hi, key);
        } else if (nx % V == 0) {
            const uint64_t lanes = nx / V;
            uint64_t gx = (lanes + B - 1) / B;
            if (gx > 64) gx = 64;
            uint64_t gy = rows < kMaxSpeedGridRows ? rows : kMaxSpeedGridRows;
            hipLaunchKernelGGL((lbm::k_max_speed<T, V, B>), dim3((unsigned)gx, (unsigned)gy), dim3(B), 0, stream,
                               (const T *)f, (uint32_t)nx, (uint32_t)ny, (uint32_t)y_lo, (uint32_t)y_hi, key);
        } else {
            uint64_t gx = (nx + B - 1) / B;
            if (gx > 64) gx = 64;
            uint64_t gy = rows < kMaxSpeedGridRows ? rows : kMaxSpeedGridRows;
            hipLaunchKernelGGL((lbm::k_max_speed<T, 1, B>), dim3((unsigned)gx, (unsigned)gy), dim3(B), 0, stream,
                               (const T *)f, (uint32_t)nx, (uint32_t)ny, (uint32_t)y_lo, (uint32_t)y_hi, key);
        }
        if (int s = launch_status("find max velocity")) return s;
    }
    hipLaunchKernelGGL((lbm::k_max_speed_finish<T>), dim3(1), dim3(1), 0, stream, (const uint32_t *)key, (T *)out_dev);
    return launch_status("max velocity finish");
}

}  // namespace

// ============================================================================================
// One kernel of the family of multi-step sweeps that the settings in force select: asking the runtime
// for its attributes makes it load the code object the whole family lives in.
template <typename T>
static const void *fused_family_member(int arith, bool aos)
{
    constexpr int V = fused_width<T>();
    if (aos) {
        if (arith == lbm::kArithContracted) return (const void *)lbm::k_fused<T, V, 1, false, false, true, lbm::kArithContracted>;
        return (const void *)lbm::k_fused<T, V, 1, false, false, true, lbm::kArithUnrolled>;
    }
    if (arith == lbm::kArithContracted) return (const void *)lbm::k_fused<T, V, 1, false, false, false, lbm::kArithContracted>;
    if (arith == lbm::kArithRolled) return (const void *)lbm::k_fused<T, V, 1, false, false, false, lbm::kArithRolled>;
    return (const void *)lbm::k_fused<T, V, 1, false, false, false, lbm::kArithUnrolled>;
}

extern "C" {

const char *lbm_status_string(int status)
{
    switch (status) {
        case LBM_OK: return "ok";
        case LBM_ERR_INVALID_ARGUMENT: return "invalid argument";
        case LBM_ERR_HIP: return "HIP runtime error";
        case LBM_ERR_NO_DEVICE: return "no usable device";
        case LBM_ERR_UNSUPPORTED: return "unsupported";
        default: return "unknown status";
    }
}

const char *lbm_last_error(void) { return g_err; }

int lbm_abi_version(void) { return 1; }

int lbm_set_layout(int layout)
{
    if (layout != LBM_LAYOUT_SOA && layout != LBM_LAYOUT_AOS)
        return fail(LBM_ERR_INVALID_ARGUMENT, "unknown layout %d", layout);
    g_layout = layout;
    return LBM_OK;
}

int lbm_get_layout(void) { return g_layout; }

// the arithmetic that UNROLL_LOOPS=true stands for: the default of this process unless that is the rolled one
int unrolled_arithmetic()
{
    const int d = arithmetic_default();
    return d == lbm::kArithRolled ? lbm::kArithContracted : d;
}

int lbm_set_unroll_loops(int unrolled)
{
    if (!unrolled) g_arith = lbm::kArithRolled;
    else if (g_arith == lbm::kArithRolled) g_arith = unrolled_arithmetic();
    return LBM_OK;
}

int lbm_get_unroll_loops(void) { return g_arith == lbm::kArithRolled ? 0 : 1; }

int lbm_set_arithmetic(int arithmetic)
{
    if (arithmetic != LBM_ARITH_UNROLLED && arithmetic != LBM_ARITH_ROLLED && arithmetic != LBM_ARITH_CONTRACTED)
        return fail(LBM_ERR_INVALID_ARGUMENT, "unknown arithmetic %d", arithmetic);
    g_arith = arithmetic;
    return LBM_OK;
}

int lbm_get_arithmetic(void) { return g_arith; }

int lbm_set_launch_shape(unsigned threads_x, unsigned waves_per_cu)
{
    if (threads_x != 0 && threads_x != 64 && threads_x != 128 && threads_x != 256 && threads_x != 512)
        return fail(LBM_ERR_INVALID_ARGUMENT, "threads_x must be 0, 64, 128, 256 or 512 (got %u)", threads_x);
    if (waves_per_cu > 32)
        return fail(LBM_ERR_INVALID_ARGUMENT, "waves_per_cu must be 0..32 (got %u)", waves_per_cu);
    g_threads_x = threads_x;
    g_waves_per_cu = waves_per_cu;
    return LBM_OK;
}

int lbm_set_tuning(const char *name, int value)
{
    if (name == nullptr) return fail(LBM_ERR_INVALID_ARGUMENT, "null tuning name");
    if (strcmp(name, "fused_narrow") == 0) g_fused_narrow = value;
    else if (strcmp(name, "fused_group") == 0) g_fused_group = value;
    else if (strcmp(name, "fused_bulk") == 0) g_fused_bulk = value;
    else if (strcmp(name, "fused_max_steps") == 0) g_fused_max_steps = value;
    else if (strcmp(name, "fused_min_nodes") == 0) g_fused_min_nodes = value;
    else if (strcmp(name, "fused_lane_nodes") == 0) g_fused_lane_nodes = value;
    else return fail(LBM_ERR_INVALID_ARGUMENT, "unknown tuning switch '%s'", name);
    return LBM_OK;
}

int lbm_ctx_create(lbm_ctx **out)
{
    if (out == nullptr) return fail(LBM_ERR_INVALID_ARGUMENT, "null context result pointer");
    lbm_ctx *c = new (std::nothrow) lbm_ctx;
    if (c == nullptr) return fail(LBM_ERR_INVALID_ARGUMENT, "out of host memory");
    c->layout = cur_layout();
    c->arith = cur_arith();
    c->threads_x = cur_threads_x();
    c->waves_per_cu = cur_waves_per_cu();
    *out = c;
    return LBM_OK;
}

int lbm_ctx_destroy(lbm_ctx *ctx)
{
    if (ctx != nullptr && t_ctx == ctx) t_ctx = nullptr;
    delete ctx;
    return LBM_OK;
}

int lbm_ctx_make_current(lbm_ctx *ctx)
{
    t_ctx = ctx;
    return LBM_OK;
}

lbm_ctx *lbm_ctx_get_current(void) { return t_ctx; }

int lbm_ctx_set_layout(lbm_ctx *ctx, int layout)
{
    if (ctx == nullptr) return fail(LBM_ERR_INVALID_ARGUMENT, "null context");
    if (layout != LBM_LAYOUT_SOA && layout != LBM_LAYOUT_AOS)
        return fail(LBM_ERR_INVALID_ARGUMENT, "unknown layout %d", layout);
    ctx->layout = layout;
    return LBM_OK;
}

int lbm_ctx_set_arithmetic(lbm_ctx *ctx, int arithmetic)
{
    if (ctx == nullptr) return fail(LBM_ERR_INVALID_ARGUMENT, "null context");
    if (arithmetic != LBM_ARITH_UNROLLED && arithmetic != LBM_ARITH_ROLLED && arithmetic != LBM_ARITH_CONTRACTED)
        return fail(LBM_ERR_INVALID_ARGUMENT, "unknown arithmetic %d", arithmetic);
    ctx->arith = arithmetic;
    return LBM_OK;
}

int lbm_ctx_set_launch_shape(lbm_ctx *ctx, unsigned threads_x, unsigned waves_per_cu)
{
    if (ctx == nullptr) return fail(LBM_ERR_INVALID_ARGUMENT, "null context");
    if (threads_x != 0 && threads_x != 64 && threads_x != 128 && threads_x != 256 && threads_x != 512)
        return fail(LBM_ERR_INVALID_ARGUMENT, "threads_x must be 0, 64, 128, 256 or 512 (got %u)", threads_x);
    if (waves_per_cu > 32)
        return fail(LBM_ERR_INVALID_ARGUMENT, "waves_per_cu must be 0..32 (got %u)", waves_per_cu);
    ctx->threads_x = threads_x;
    ctx->waves_per_cu = waves_per_cu;
    return LBM_OK;
}

int lbm_ctx_get_layout(const lbm_ctx *ctx) { return ctx ? ctx->layout : -1; }

int lbm_ctx_get_arithmetic(const lbm_ctx *ctx) { return ctx ? ctx->arith : -1; }

int lbm_init_shearwave(void *f, int prec, uint64_t nx, uint64_t ny, double rho_init, double u,
                       uint64_t y_below, uint64_t y_glob, unsigned halo, void *stream)
{
    if (int s = check_common(f, prec, nx, ny)) return s;
    if (ny < 2ull * halo) return fail(LBM_ERR_INVALID_ARGUMENT, "ny=%llu too small for halo", (unsigned long long)ny);
    if (y_glob == 0) return fail(LBM_ERR_INVALID_ARGUMENT, "y_glob must be positive");
    hipStream_t st = (hipStream_t)stream;
    return prec == LBM_F32 ? init_shearwave_t<float>(f, nx, ny, rho_init, u, y_below, y_glob, halo, st)
                           : init_shearwave_t<double>(f, nx, ny, rho_init, u, y_below, y_glob, halo, st);
}

int lbm_init_rest(void *f, int prec, uint64_t nx, uint64_t ny, double rho_init, void *stream)
{
    if (int s = check_common(f, prec, nx, ny)) return s;
    hipStream_t st = (hipStream_t)stream;
    return prec == LBM_F32 ? init_rest_t<float>(f, nx, ny, rho_init, st)
                           : init_rest_t<double>(f, nx, ny, rho_init, st);
}

int lbm_pull_periodic_no_mpi(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om, void *stream)
{
    return periodic_rows<true>(f, buf, prec, nx, ny, om, 0, ny, 1, stream);
}
int lbm_push_periodic_no_mpi(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om, void *stream)
{
    return periodic_rows<false>(f, buf, prec, nx, ny, om, 0, ny, 1, stream);
}
int lbm_pull_periodic_rows(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                           uint64_t y_lo, uint64_t y_hi, int wrap_y, void *stream)
{
    return periodic_rows<true>(f, buf, prec, nx, ny, om, y_lo, y_hi, wrap_y, stream);
}
int lbm_push_periodic_rows(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                           uint64_t y_lo, uint64_t y_hi, int wrap_y, void *stream)
{
    return periodic_rows<false>(f, buf, prec, nx, ny, om, y_lo, y_hi, wrap_y, stream);
}
int lbm_pull_periodic_inner(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                            uint64_t LX, uint64_t LY, uint64_t HX, uint64_t HY, void *stream)
{
    return inner<true>(f, buf, prec, nx, ny, om, LX, LY, HX, HY, stream);
}
int lbm_push_periodic_inner(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                            uint64_t LX, uint64_t LY, uint64_t HX, uint64_t HY, void *stream)
{
    return inner<false>(f, buf, prec, nx, ny, om, LX, LY, HX, HY, stream);
}
int lbm_pull_periodic_outer(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om, void *stream)
{
    return outer<true>(f, buf, prec, nx, ny, om, stream);
}
int lbm_push_periodic_outer(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om, void *stream)
{
    return outer<false>(f, buf, prec, nx, ny, om, stream);
}

int lbm_push_lid_driven(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                        double rho_eq, double u_lid, void *stream)
{
    return lid_rows<false>(f, buf, prec, nx, ny, om, rho_eq, u_lid, 0, ny, 0, ny, stream);
}
int lbm_pull_lid_driven(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                        double rho_eq, double u_lid, void *stream)
{
    return lid_rows<true>(f, buf, prec, nx, ny, om, rho_eq, u_lid, 0, ny, 0, ny, stream);
}
int lbm_push_lid_rows(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                      double rho_eq, double u_lid, uint64_t y_lo, uint64_t y_hi, int64_t y_goff,
                      uint64_t ny_glob, void *stream)
{
    return lid_rows<false>(f, buf, prec, nx, ny, om, rho_eq, u_lid, y_lo, y_hi, y_goff, ny_glob, stream);
}
int lbm_pull_lid_rows(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                      double rho_eq, double u_lid, uint64_t y_lo, uint64_t y_hi, int64_t y_goff,
                      uint64_t ny_glob, void *stream)
{
    return lid_rows<true>(f, buf, prec, nx, ny, om, rho_eq, u_lid, y_lo, y_hi, y_goff, ny_glob, stream);
}

int lbm_load_kernels(int prec)
{
    if (prec != LBM_F32 && prec != LBM_F64) return fail(LBM_ERR_INVALID_ARGUMENT, "precision tag must be 4 or 8 (got %d)", prec);
    const bool aos = cur_layout() == LBM_LAYOUT_AOS;
    const int arith = cur_arith();
    const void *members[2] = {
        prec == LBM_F32 ? fused_family_member<float>(arith, aos) : fused_family_member<double>(arith, aos),
        // the one-step sweeps, the initialisers and the output kernels share a code object
        prec == LBM_F32 ? (const void *)lbm::k_max_speed<float, 1, 256> : (const void *)lbm::k_max_speed<double, 1, 256>,
    };
    for (const void *k : members) {
        hipFuncAttributes attr;
        const hipError_t e = hipFuncGetAttributes(&attr, k);
        if (e != hipSuccess) return fail(LBM_ERR_HIP, "loading the sweep kernels: %s", hipGetErrorString(e));
    }
    return LBM_OK;
}

int lbm_fused_max_steps(int prec, uint64_t nx)
{
    if (prec != LBM_F32 && prec != LBM_F64) return 0;
    bool ok = prec == LBM_F32 ? fused_supported<float>(nx) : fused_supported<double>(nx);
    // the rolled and contracted arithmetics are built for 8-byte lanes only, the rolled one not for
    // [NY][NX][9] arrays
    if (cur_arith() != lbm::kArithUnrolled && prec == LBM_F32 && (nx % 2 != 0 || nx < 4)) ok = false;
    if (cur_arith() == lbm::kArithRolled && cur_layout() == LBM_LAYOUT_AOS) ok = false;
    return ok ? max_fused_steps(prec) : 1;
}
int lbm_plan_steps(int prec, uint64_t nx, uint64_t rows, uint64_t remaining)
{
    return lbm_plan_steps_scene(prec, nx, rows, remaining, 0);
}
int lbm_plan_steps_scene(int prec, uint64_t nx, uint64_t rows, uint64_t remaining, int lid)
{
    if (remaining == 0) return 0;
    const int kmax = lbm_fused_max_steps(prec, nx);
    if (kmax <= 1) return kmax;   // 0: bad precision, 1: one-step sweeps only
    // Measured on MI355X (fp32 pull, square lattices, one step per launch -> four): 512^2 23 -> 38
    // GLUPS, 1024^2 65 -> 95, 2048^2 68 -> 137, 4096^2 86 -> 180, 8192^2 77 -> 231, 16384^2 85 -> 242:
    // fusing pays at every size (small lattices are launch-bound, large ones HBM-bound), so there is
    // no size threshold by default; LBM_FUSED_MIN_NODES sets one.
    const int min_nodes = g_fused_min_nodes.load();
    if (min_nodes > 0 && nx * rows < (uint64_t)min_nodes) return 1;
    // Large lattices (16384^2, GLUPS at 6/7/8 steps): fp32 strict 325/334/335, contracted 377/410/424
    // (push 372/401/411); fp64 contracted pull 162/167/171, push 159/166/162.  The strict and rolled
    // spellings and the fp64 sweeps take seven (an eighth step gains little or nothing and deepens
    // the halos of a decomposed run), contracted fp32 eight.  The cavity is planned like the periodic
    // scene: all but its outermost wavefronts run the periodic sweep's code (16384^2 fp32
    // contracted, seven / eight steps: pull 391 / 416 GLUPS, push 397 / 412); `lid` is kept for
    // kernels that may differ again.
    (void)lid;
    uint64_t dflt = (uint64_t)kmax;
    if (dflt > 7 && (cur_arith() != lbm::kArithContracted || prec == LBM_F64)) dflt = 7;
    // Small lattices (one round of wavefronts or less) are bound by the chain of dependent iterations
    // of a wavefront; more steps lengthen the chain.  fp32 pull, GLUPS at 3/4/5/6/7/8 steps: 512^2
    // 50/56/57/53/52/49, 1024^2 107/128/144/137/136/133, 1536^2 135/160/197/203/212/213, 2048^2
    // 161/196/225/233/253/260.
    if (dflt > (uint64_t)lbm::kFusedNarrowSteps && nx * rows < kSmallLatticeNodes) dflt = (uint64_t)lbm::kFusedNarrowSteps;
    const int forced = g_fused_max_steps.load();   // LBM_FUSED_MAX_STEPS / lbm_set_tuning
    uint64_t k = forced > 0 ? (uint64_t)forced : dflt;
    if (k < 1) k = 1;
    if (k > (uint64_t)kmax) k = (uint64_t)kmax;
    // The run is cut into the fewest launches of at most k steps, and those launches are made as
    // equal as possible: 20 steps are 5+5+5+5, not 6+6+6+2 -- a short tail launch pays a whole pass
    // over memory for its few steps (16384^2 fp32: two steps per launch run at half the rate of six).
    const uint64_t launches = (remaining + k - 1) / k;
    return (int)((remaining + launches - 1) / launches);
}
int lbm_pull_periodic_fused(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                            uint64_t y_lo, uint64_t y_hi, int wrap_y, int nsteps, void *stream)
{
    return periodic_fused<true>(f, buf, prec, nx, ny, om, y_lo, y_hi, wrap_y, nsteps, stream);
}
int lbm_push_periodic_fused(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                            uint64_t y_lo, uint64_t y_hi, int wrap_y, int nsteps, void *stream)
{
    return periodic_fused<false>(f, buf, prec, nx, ny, om, y_lo, y_hi, wrap_y, nsteps, stream);
}
int lbm_push_lid_fused(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                       double rho_eq, double u_lid, uint64_t y_lo, uint64_t y_hi, int64_t y_goff,
                       uint64_t ny_glob, int nsteps, void *stream)
{
    return lid_fused<false>(f, buf, prec, nx, ny, om, rho_eq, u_lid, y_lo, y_hi, y_goff, ny_glob, nsteps, stream);
}
int lbm_pull_lid_fused(const void *f, void *buf, int prec, uint64_t nx, uint64_t ny, double om,
                       double rho_eq, double u_lid, uint64_t y_lo, uint64_t y_hi, int64_t y_goff,
                       uint64_t ny_glob, int nsteps, void *stream)
{
    return lid_fused<true>(f, buf, prec, nx, ny, om, rho_eq, u_lid, y_lo, y_hi, y_goff, ny_glob, nsteps, stream);
}

int lbm_compute_velocities(const void *f, void *vel, int prec, uint64_t nx, uint64_t ny,
                           uint64_t y_lo, uint64_t y_hi, void *stream)
{
    if (int s = check_common(f, prec, nx, ny)) return s;
    if (int s = check_rows(ny, y_lo, y_hi, false)) return s;
    if (vel == nullptr) return fail(LBM_ERR_INVALID_ARGUMENT, "null velocity pointer");
    hipStream_t st = (hipStream_t)stream;
    return prec == LBM_F32 ? velocity_t<float, 0>(f, vel, nx, ny, y_lo, y_hi, st)
                           : velocity_t<double, 0>(f, vel, nx, ny, y_lo, y_hi, st);
}

int lbm_velocity_magnitudes(const void *f, void *mag, int prec, uint64_t nx, uint64_t ny,
                            uint64_t y_lo, uint64_t y_hi, void *stream)
{
    if (int s = check_common(f, prec, nx, ny)) return s;
    if (int s = check_rows(ny, y_lo, y_hi, false)) return s;
    if (mag == nullptr) return fail(LBM_ERR_INVALID_ARGUMENT, "null magnitude pointer");
    hipStream_t st = (hipStream_t)stream;
    return prec == LBM_F32 ? velocity_t<float, 1>(f, mag, nx, ny, y_lo, y_hi, st)
                           : velocity_t<double, 1>(f, mag, nx, ny, y_lo, y_hi, st);
}

int lbm_max_velocity_dev(const void *f, void *out_dev, int prec, uint64_t nx, uint64_t ny,
                         uint64_t y_lo, uint64_t y_hi, void *stream)
{
    if (int s = check_common(f, prec, nx, ny)) return s;
    if (int s = check_rows(ny, y_lo, y_hi, false)) return s;
    if (out_dev == nullptr || (uintptr_t)out_dev % 8 != 0)
        return fail(LBM_ERR_INVALID_ARGUMENT, "result pointer must be non-null and 8-byte aligned");
    // the caller's result word doubles as the reduction's key (cleared, raised by atomicMax, decoded in
    // place): no state shared between calls, nothing allocated
    uint32_t *key = (uint32_t *)out_dev;
    hipStream_t st = (hipStream_t)stream;
    return prec == LBM_F32 ? max_velocity_t<float>(f, out_dev, nx, ny, y_lo, y_hi, key, sizeof(float), st)
                           : max_velocity_t<double>(f, out_dev, nx, ny, y_lo, y_hi, key, sizeof(double), st);
}

int lbm_max_velocity_dev_static(const void *f, void *out_dev, int prec, uint64_t nx, uint64_t ny,
                                uint64_t y_lo, uint64_t y_hi, void *stream)
{
    if (int s = check_common(f, prec, nx, ny)) return s;
    if (int s = check_rows(ny, y_lo, y_hi, false)) return s;
    if (out_dev == nullptr || (uintptr_t)out_dev % 8 != 0)
        return fail(LBM_ERR_INVALID_ARGUMENT, "result pointer must be non-null and 8-byte aligned");
    // the reduction needs one key word whatever the lattice size; the static array holds one line
    static_assert(kStaticScratchBytes >= sizeof(uint32_t), "static scratch smaller than the reduction's key");
    uint32_t *key = nullptr;
    if (int s = static_scratch(&key)) return s;
    hipStream_t st = (hipStream_t)stream;
    // the array keeps the previous call's key: the whole line is cleared on the stream first
    return prec == LBM_F32 ? max_velocity_t<float>(f, out_dev, nx, ny, y_lo, y_hi, key, kStaticScratchBytes, st)
                           : max_velocity_t<double>(f, out_dev, nx, ny, y_lo, y_hi, key, kStaticScratchBytes, st);
}

int lbm_max_velocity(const void *f, int prec, uint64_t nx, uint64_t ny, uint64_t y_lo,
                     uint64_t y_hi, void *stream, double *out_host)
{
    if (out_host == nullptr) return fail(LBM_ERR_INVALID_ARGUMENT, "null result pointer");
    if (int s = check_common(f, prec, nx, ny)) return s;
    if (int s = check_rows(ny, y_lo, y_hi, false)) return s;
    uint32_t *key = nullptr;
    if (int s = key_scratch(&key)) return s;
    // the decoded value goes to the second half of the scratch line
    void *res = (void *)((char *)key + 128);
    hipStream_t st = (hipStream_t)stream;
    int s = prec == LBM_F32 ? max_velocity_t<float>(f, res, nx, ny, y_lo, y_hi, key, sizeof(uint32_t), st)
                            : max_velocity_t<double>(f, res, nx, ny, y_lo, y_hi, key, sizeof(uint32_t), st);
    if (s) return s;
    double host = 0;
    HIP_TRY(hipMemcpyAsync(&host, res, (size_t)prec, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (prec == LBM_F32) {
        float v;
        memcpy(&v, &host, sizeof v);
        *out_host = (double)v;
    } else {
        *out_host = host;
    }
    return LBM_OK;
}

int lbm_selftest_reciprocal(const void *x, void *out, int prec, uint64_t n, void *stream)
{
    if (prec != LBM_F32 && prec != LBM_F64) return fail(LBM_ERR_INVALID_ARGUMENT, "precision tag %d is neither 4 nor 8", prec);
    if (n == 0) return LBM_OK;
    if (x == nullptr || out == nullptr) return fail(LBM_ERR_INVALID_ARGUMENT, "null pointer");
    const uint64_t blocks = (n + 255) / 256;
    const dim3 grid((unsigned)(blocks < 65536 ? blocks : 65536)), block(256);
    if (prec == LBM_F32)
        hipLaunchKernelGGL((lbm::k_reciprocal<float>), grid, block, 0, (hipStream_t)stream, (const float *)x, (float *)out, n);
    else
        hipLaunchKernelGGL((lbm::k_reciprocal<double>), grid, block, 0, (hipStream_t)stream, (const double *)x, (double *)out, n);
    return launch_status("reciprocal self-check");
}

}  // extern "C"
