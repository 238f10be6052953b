// Device code of the D2Q9 BGK collide-and-stream path for gfx950 (MI355X, wave64).
//
// Design: the sweep is a pure HBM stream -- every population is read exactly once and written
// exactly once per step (9 reads + 9 writes per node, ~110 flops) -- so the kernels are shaped
// around memory instructions, not arithmetic:
//   * SoA [9][NY][NX]; one lane owns V consecutive x of one row, V*sizeof(T) = 16 bytes, so every
//     access is a 128-bit global_load/store_dwordx4 and a wave moves 1 KiB per instruction;
//   * the three "x-1" and three "x+1" channels are fetched (pull) or written (push) with 4-byte
//     aligned 128-bit accesses, shifted by one element; only the first and last lane of a row take
//     the element-wise wrap path;
//   * row base addresses are wave-uniform (SGPR), the per-lane part is one 32-bit x offset;
//   * one wave handles one 1 KiB chunk of one row, straight-line (no loop): measured on MI355X the
//     chip streams fastest with a huge grid of short waves dispatched in address order, about
//     8-12 resident waves per CU (more bytes in flight lower the achieved HBM rate), and
//     non-temporal stores for the write stream (+7 %); the host caps residency with an otherwise
//     unused dynamic-LDS allocation;
//   * all scalars (nx, ny, omega, ...) are kernel arguments;
//   * arithmetic follows the reference's expression trees with FMA contraction off, so results
//     are bit-identical to a strict IEEE evaluation of the reference source.
//
// Reference citations are to JulianKarrer/lbm src/ (pull-push.cpp, init.cpp, io.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace lbm {

constexpr int kWave = 64;

// ------------------------------------------------------------------------------------------
// 128-bit vector access helpers.  VecA: naturally aligned; VecU: element aligned (shifted rows).
template <typename T, int V>
struct VecTypes {
    typedef T vec_t __attribute__((ext_vector_type(V)));
    typedef vec_t vec_a __attribute__((aligned(sizeof(T) * V)));
    typedef vec_t vec_u __attribute__((aligned(sizeof(T))));
};

template <typename T, int V>
__device__ __forceinline__ void load_a(T (&r)[V], const T *p)
{
    typename VecTypes<T, V>::vec_t v = *reinterpret_cast<const typename VecTypes<T, V>::vec_a *>(p);
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = v[i];
}
template <typename T, int V>
__device__ __forceinline__ void load_u(T (&r)[V], const T *p)
{
    typename VecTypes<T, V>::vec_t v = *reinterpret_cast<const typename VecTypes<T, V>::vec_u *>(p);
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = v[i];
}
// Stores are non-temporal: the write stream is never re-read within a sweep.
template <typename T, int V>
__device__ __forceinline__ void store_a(T *p, const T (&r)[V])
{
    typename VecTypes<T, V>::vec_t v;
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = r[i];
    __builtin_nontemporal_store(v, reinterpret_cast<typename VecTypes<T, V>::vec_a *>(p));
}
template <typename T, int V>
__device__ __forceinline__ void store_u(T *p, const T (&r)[V])
{
    typename VecTypes<T, V>::vec_t v;
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = r[i];
    __builtin_nontemporal_store(v, reinterpret_cast<typename VecTypes<T, V>::vec_u *>(p));
}

// r[i] = row[(x0 + i - 1) mod nx]   (values arriving from the left neighbour, periodic in x)
template <typename T, int V>
__device__ __forceinline__ void load_from_left(T (&r)[V], const T *row, uint32_t x0, uint32_t nx)
{
    if (x0 != 0) {
        load_u<T, V>(r, row + (x0 - 1));
    } else {
        r[0] = row[nx - 1];
#pragma unroll
        for (int i = 1; i < V; ++i) r[i] = row[i - 1];
    }
}
// r[i] = row[(x0 + i + 1) mod nx]   (values arriving from the right neighbour)
template <typename T, int V>
__device__ __forceinline__ void load_from_right(T (&r)[V], const T *row, uint32_t x0, uint32_t nx)
{
    if (x0 + V != nx) {
        load_u<T, V>(r, row + (x0 + 1));
    } else {
#pragma unroll
        for (int i = 0; i < V - 1; ++i) r[i] = row[x0 + 1 + i];
        r[V - 1] = row[0];
    }
}
// row[(x0 + i + 1) mod nx] = r[i]   (values leaving to the right neighbour)
template <typename T, int V>
__device__ __forceinline__ void store_to_right(T *row, uint32_t x0, uint32_t nx, const T (&r)[V])
{
    if (x0 + V != nx) {
        store_u<T, V>(row + (x0 + 1), r);
    } else {
#pragma unroll
        for (int i = 0; i < V - 1; ++i) row[x0 + 1 + i] = r[i];
        row[0] = r[V - 1];
    }
}
// row[(x0 + i - 1) mod nx] = r[i]   (values leaving to the left neighbour)
template <typename T, int V>
__device__ __forceinline__ void store_to_left(T *row, uint32_t x0, uint32_t nx, const T (&r)[V])
{
    if (x0 != 0) {
        store_u<T, V>(row + (x0 - 1), r);
    } else {
        row[nx - 1] = r[0];
#pragma unroll
        for (int i = 1; i < V; ++i) row[i - 1] = r[i];
    }
}

// ------------------------------------------------------------------------------------------
// BGK relaxation of one node.  Expression trees as in pull-push.cpp:25-72 (pull), :138-185
// (push), :454-483 (lid): rho summed left to right, one division, shared sub-expressions
// f45*s*s = (f45*s)*s, bracket ((1 -+ 3a) + a2) - usq, out = f + om*((w*bracket) - f).
// Constants are rounded once from double (pull-push.cpp:39-43).
// X is the arithmetic type of one "lane group": T itself, or a short vector of T whose elements are
// independent nodes (elementwise IEEE operations, so every node sees exactly the scalar expression
// tree).  With X = float x 2 the adds, subtracts and multiplies become packed v_pk_* instructions.
template <typename X, typename T>
__device__ __forceinline__ void collide_x(const X f_0, const X f_1, const X f_2, const X f_3,
                                          const X f_4, const X f_5, const X f_6, const X f_7,
                                          const X f_8, const T om_, X (&o)[9])
{
    const X om = X(om_);
    const X rho = f_0 + f_1 + f_2 + f_3 + f_4 + f_5 + f_6 + f_7 + f_8;
    const X rho_inv = X(T(1.0)) / rho;
    const X ux = (f_1 - f_3 + f_5 - f_6 - f_7 + f_8) * rho_inv;
    const X uy = (f_2 - f_4 + f_5 + f_6 - f_7 - f_8) * rho_inv;
    const X f45 = X(T(4.5)), f15 = X(T(1.5));
    const X f49 = X(T(4. / 9.)), f19 = X(T(1. / 9.)), f136 = X(T(1. / 36.));
    const X one = X(T(1)), three = X(T(3));
    const X w_1_36 = rho * f136;
    const X w_1_9 = rho * f19;
    const X w_4_9 = rho * f49;
    const X uxpuy = ux + uy;
    const X uxpuy_2 = f45 * uxpuy * uxpuy;
    const X uymux = uy - ux;
    const X uymux_2 = f45 * uymux * uymux;
    const X ux_2 = f45 * ux * ux;
    const X uy_2 = f45 * uy * uy;
    const X usq = f15 * (ux * ux + uy * uy);
    o[7] = f_7 + om * ((w_1_36 * (one - three * uxpuy + uxpuy_2 - usq)) - f_7);
    o[4] = f_4 + om * ((w_1_9 * (one - three * uy + uy_2 - usq)) - f_4);
    o[8] = f_8 + om * ((w_1_36 * (one - three * uymux + uymux_2 - usq)) - f_8);
    o[3] = f_3 + om * ((w_1_9 * (one - three * ux + ux_2 - usq)) - f_3);
    o[0] = f_0 + om * ((w_4_9 * (one - usq)) - f_0);
    o[1] = f_1 + om * ((w_1_9 * (one + three * ux + ux_2 - usq)) - f_1);
    o[6] = f_6 + om * ((w_1_36 * (one + three * uymux + uymux_2 - usq)) - f_6);
    o[2] = f_2 + om * ((w_1_9 * (one + three * uy + uy_2 - usq)) - f_2);
    o[5] = f_5 + om * ((w_1_36 * (one + three * uxpuy + uxpuy_2 - usq)) - f_5);
}

// The reference's other spelling of the same relaxation, compiled there when UNROLL_LOOPS is false
// (pull_periodic_for / push_periodic_for, pull-push.cpp:90-119, :204-231): loops over the nine
// directions with the lattice vectors and weights in tables.  Density and velocity come out the same
// (adding 0 and multiplying by 0 or +-1 change nothing that is finite); the equilibrium is
// (w*rho) * (((1 + 3c) + 4.5*(c*c)) - 1.5*u2) with c = cx*ux + cy*uy, where the unrolled form has
// (4.5*c)*c -- a different rounding, so the two spellings differ in the last bits and each is
// followed literally.
template <typename X, typename T>
__device__ __forceinline__ void collide_for_x(const X f_0, const X f_1, const X f_2, const X f_3,
                                              const X f_4, const X f_5, const X f_6, const X f_7,
                                              const X f_8, const T om_, X (&o)[9])
{
    const X om = X(om_);
    const X f[9] = {f_0, f_1, f_2, f_3, f_4, f_5, f_6, f_7, f_8};
    X rho = X(T(0.));
#pragma unroll
    for (int i = 0; i < 9; ++i) rho = rho + f[i];
    const X rho_inv = X(T(1.0)) / rho;
    const T cx[9] = {0, 1, 0, -1, 0, 1, -1, -1, 1};
    const T cy[9] = {0, 0, 1, 0, -1, 1, 1, -1, -1};
    X ux = X(T(0.)), uy = X(T(0.));
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        ux = ux + f[i] * X(cx[i]);
        uy = uy + f[i] * X(cy[i]);
    }
    ux = ux * rho_inv;
    uy = uy * rho_inv;
    const T w[9] = {T(4. / 9.), T(1. / 9.), T(1. / 9.), T(1. / 9.), T(1. / 9.),
                    T(1. / 36.), T(1. / 36.), T(1. / 36.), T(1. / 36.)};
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const X cx_ux = X(cx[i]) * ux;
        const X cy_uy = X(cy[i]) * uy;
        const X c_i_u_r = cx_ux + cy_uy;
        const X u_2 = ux * ux + uy * uy;
        const X f_eq = X(w[i]) * rho * (X(T(1)) + X(T(3)) * c_i_u_r + X(T(4.5)) * (c_i_u_r * c_i_u_r) - X(T(1.5)) * u_2);
        o[i] = f[i] + om * (f_eq - f[i]);
    }
}

// Spellings of the relaxation (template parameter ARITH of everything below).
constexpr int kArithUnrolled = 0;     // collide_x: pull_periodic / push_periodic / push_lid_driven
constexpr int kArithRolled = 1;       // collide_for_x: the *_for functions (periodic sweeps only)
constexpr int kArithContracted = 2;   // collide_fma_x: explicit fused multiply-adds (all sweeps)

// 1/x for fp32, correctly rounded like the IEEE division it stands for, in three instructions
// instead of the eleven of the division sequence: v_rcp_f32 is good to one ulp, and one Newton step
// in fused arithmetic, r + r*(1 - x*r), then rounds correctly (Markstein's theorem; checked
// exhaustively on the device over every fp32 in [0.5, 4): tests/test_contracted_gpu.py).  Valid
// while x and 1/x are normal numbers (2^-126 <= |x| <= 2^126): densities of any physical state.
__device__ __forceinline__ float recip_exact(const float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
// 1/x for fp64: the operations of the compiler's own expansion of an IEEE division with the numerator
// 1 -- hardware estimate, two Newton steps, the final correction of the quotient, all fused -- without
// its range scaling and fix-up (v_div_scale x 2, the multiplication by the numerator, v_div_fixup),
// which leave the operands alone while x and 1/x are normal numbers: seven instructions for eleven,
// the same bits (tests/test_contracted_gpu.py compares with the division on the host).
__device__ __forceinline__ double recip_exact(const double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(e, r, r);
}
template <typename X, typename T>
__device__ __forceinline__ X recip_x(const X x)
{
    if constexpr (sizeof(T) == 8 && sizeof(X) == sizeof(T)) {
        return X(recip_exact(x));
    } else if constexpr (sizeof(T) == 8) {
        return X(T(1.0)) / x;
    } else if constexpr (sizeof(X) == sizeof(T)) {
        return X(recip_exact(x));
    } else {
        X r;
        r.x = recip_exact(x.x);
        r.y = recip_exact(x.y);
        return r;
    }
}

// One contraction of collide_x that a compiler may apply to the reference's text (the reference is
// built -O3 -march=native, CMakeLists.txt:18, where a*b+c may fuse; which ones do differs from build
// to build).  This spelling fixes one choice with explicit fused multiply-adds, so it does not depend
// on compiler flags and the CPU checker can restate it operation by operation:
//     ux*ux + uy*uy -> fma(ux, ux, uy*uy)          1 -+ 3*a        -> fma(-+3, a, 1)
//     ... + (f45*a)*a -> fma(f45*a, a, ...)        (w*bracket) - f -> fma(w, bracket, -f)
//     f + om*(...)  -> fma(om, ..., f)
// Sums, the reciprocal of the density and all other products round as in collide_x.  A third fewer
// vector instructions per node; populations stay within 1e-6 (fp32) / 1e-15 (fp64) of collide_x's
// over thousands of steps.
template <typename X, typename T>
__device__ __forceinline__ void collide_fma_x(const X f_0, const X f_1, const X f_2, const X f_3,
                                              const X f_4, const X f_5, const X f_6, const X f_7,
                                              const X f_8, const T om_, X (&o)[9])
{
    const X om = X(om_);
    const X rho = f_0 + f_1 + f_2 + f_3 + f_4 + f_5 + f_6 + f_7 + f_8;
    const X rho_inv = recip_x<X, T>(rho);
    const X ux = (f_1 - f_3 + f_5 - f_6 - f_7 + f_8) * rho_inv;
    const X uy = (f_2 - f_4 + f_5 + f_6 - f_7 - f_8) * rho_inv;
    const X f45 = X(T(4.5)), f15 = X(T(1.5));
    const X f49 = X(T(4. / 9.)), f19 = X(T(1. / 9.)), f136 = X(T(1. / 36.));
    const X one = X(T(1)), three = X(T(3)), mthree = X(T(-3));
    const X w_1_36 = rho * f136;
    const X w_1_9 = rho * f19;
    const X w_4_9 = rho * f49;
    const X uxpuy = ux + uy;
    const X uymux = uy - ux;
    const X p_s = f45 * uxpuy, p_d = f45 * uymux, p_x = f45 * ux, p_y = f45 * uy;
    const X usq = f15 * __builtin_elementwise_fma(ux, ux, uy * uy);
    auto relax = [&](const X f, const X w, const X sign3, const X a, const X p) -> X {
        const X bracket = __builtin_elementwise_fma(p, a, __builtin_elementwise_fma(sign3, a, one)) - usq;
        return __builtin_elementwise_fma(om, __builtin_elementwise_fma(w, bracket, -f), f);
    };
    o[7] = relax(f_7, w_1_36, mthree, uxpuy, p_s);
    o[4] = relax(f_4, w_1_9, mthree, uy, p_y);
    o[8] = relax(f_8, w_1_36, mthree, uymux, p_d);
    o[3] = relax(f_3, w_1_9, mthree, ux, p_x);
    o[0] = __builtin_elementwise_fma(om, __builtin_elementwise_fma(w_4_9, one - usq, -f_0), f_0);
    o[1] = relax(f_1, w_1_9, three, ux, p_x);
    o[6] = relax(f_6, w_1_36, three, uymux, p_d);
    o[2] = relax(f_2, w_1_9, three, uy, p_y);
    o[5] = relax(f_5, w_1_36, three, uxpuy, p_s);
}

template <typename X, typename T, int ARITH>
__device__ __forceinline__ void collide_arith(const X f_0, const X f_1, const X f_2, const X f_3,
                                              const X f_4, const X f_5, const X f_6, const X f_7,
                                              const X f_8, const T om, X (&o)[9])
{
    static_assert(ARITH >= kArithUnrolled && ARITH <= kArithContracted, "unknown arithmetic");
    if constexpr (ARITH == kArithRolled)          collide_for_x<X, T>(f_0, f_1, f_2, f_3, f_4, f_5, f_6, f_7, f_8, om, o);
    else if constexpr (ARITH == kArithContracted) collide_fma_x<X, T>(f_0, f_1, f_2, f_3, f_4, f_5, f_6, f_7, f_8, om, o);
    else                                          collide_x<X, T>(f_0, f_1, f_2, f_3, f_4, f_5, f_6, f_7, f_8, om, o);
}

template <typename T, int ARITH = kArithUnrolled>
__device__ __forceinline__ void collide(const T f_0, const T f_1, const T f_2, const T f_3,
                                        const T f_4, const T f_5, const T f_6, const T f_7,
                                        const T f_8, const T om, T (&o)[9])
{
    collide_arith<T, T, ARITH>(f_0, f_1, f_2, f_3, f_4, f_5, f_6, f_7, f_8, om, o);
}

// Relax the V nodes of one lane: c[q][i] -> o[q][i].  fp32 nodes go through the arithmetic in
// pairs (packed instructions), everything else one by one.
template <typename T, int V, bool PAIRS = true, int ARITH = kArithUnrolled>
__device__ __forceinline__ void collide_nodes(T (&o)[9][V], const T (&c)[9][V], const T om)
{
    if constexpr (PAIRS && sizeof(T) == 4 && V % 2 == 0) {
        typedef T pair_t __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < V; i += 2) {
            pair_t in[9], r[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) { in[q].x = c[q][i]; in[q].y = c[q][i + 1]; }
            collide_arith<pair_t, T, ARITH>(in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7], in[8], om, r);
#pragma unroll
            for (int q = 0; q < 9; ++q) { o[q][i] = r[q].x; o[q][i + 1] = r[q].y; }
        }
    } else {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            T r[9];
            collide<T, ARITH>(c[0][i], c[1][i], c[2][i], c[3][i], c[4][i], c[5][i], c[6][i], c[7][i], c[8][i], om, r);
#pragma unroll
            for (int q = 0; q < 9; ++q) o[q][i] = r[q];
        }
    }
}

// Macroscopic velocity of one node as io.cpp:143-171 and io.cpp:190-202 form it (the two sites
// give identical values: (sum)*rho_inv).
template <typename T>
__device__ __forceinline__ void node_velocity(const T f_0, const T f_1, const T f_2, const T f_3,
                                              const T f_4, const T f_5, const T f_6, const T f_7,
                                              const T f_8, T &ux, T &uy)
{
    const T sx = f_1 - f_3 + f_5 - f_6 - f_7 + f_8;
    const T sy = f_2 - f_4 + f_5 + f_6 - f_7 - f_8;
    const T rho_inv = T(1.0) / (f_0 + f_1 + f_2 + f_3 + f_4 + f_5 + f_6 + f_7 + f_8);
    ux = sx * rho_inv;
    uy = sy * rho_inv;
}
// |u| as io.cpp:204 / :238: SQRT is sqrtf for both precisions (global.h:30).
template <typename T>
__device__ __forceinline__ T node_speed(const T ux, const T uy)
{
    return (T)sqrtf((float)(ux * ux + uy * uy));
}

// ------------------------------------------------------------------------------------------
// Geometry shared by all sweeps.  Array is [9][ny][nx]; plane = ny*nx elements.
struct Geom {
    int xcd_rows;     // != 0: give each XCD whole rows (see block_coords)
    uint32_t nx;      // row length
    uint32_t ny;      // rows in the array (incl. ghost rows if any)
    uint32_t y_lo;    // first row to update
    uint32_t y_hi;    // one past the last row to update
    int wrap_y;       // periodic y inside the array
    // lid scenes: global row = row + y_goff, cavity height ny_glob
    int64_t y_goff;
    uint64_t ny_glob;
};

template <typename T>
__device__ __forceinline__ const T *rowp(const T *base, size_t plane, int q, uint32_t y, uint32_t nx)
{
    return base + ((size_t)q * plane + (size_t)y * nx);
}
template <typename T>
__device__ __forceinline__ T *rowp(T *base, size_t plane, int q, uint32_t y, uint32_t nx)
{
    return base + ((size_t)q * plane + (size_t)y * nx);
}

// Row handled by this workgroup: grid = (x chunks, rows mod kRowsPerGridZ, rows / kRowsPerGridZ).
constexpr uint32_t kRowsPerGridZ = 32768;
__device__ __forceinline__ uint32_t block_row(const Geom &g)
{
    return g.y_lo + blockIdx.z * kRowsPerGridZ + blockIdx.y;
}

// (x chunk, row) of this workgroup for the vector sweeps.  Workgroups are dealt to the 8 XCDs
// round-robin in linear order (x fastest), so with the plain mapping the two x-neighbours of a
// chunk always run on other XCDs and the cache line that a shifted 1 KiB access shares with the
// neighbouring chunk is fetched from memory once per XCD (measured: +8 % FETCH_SIZE).  With
// xcd_rows set, each group of 8 consecutive rows is re-dealt so that one XCD walks one whole row:
// linear id r inside the group -> row r % 8, chunk r / 8.  Placement only affects speed, never
// results.  Rows beyond the last full group of 8 keep the plain mapping.
__device__ __forceinline__ void block_coords(const Geom &g, uint32_t &chunk, uint32_t &y)
{
    uint32_t bx = blockIdx.x;
    uint32_t by = blockIdx.z * kRowsPerGridZ + blockIdx.y;  // row index relative to y_lo
    const uint32_t rows = g.y_hi - g.y_lo;
    if (g.xcd_rows && by < (rows & ~7u)) {
        const uint32_t r = (by & 7u) * gridDim.x + bx;
        by = (by & ~7u) + (r & 7u);
        bx = r >> 3;
    }
    chunk = bx;
    y = g.y_lo + by;
}

__device__ __forceinline__ uint32_t row_up(const Geom &g, uint32_t y)
{
    return g.wrap_y ? ((y + 1 == g.ny) ? 0u : y + 1) : y + 1;
}
__device__ __forceinline__ uint32_t row_down(const Geom &g, uint32_t y)
{
    return g.wrap_y ? ((y == 0) ? g.ny - 1 : y - 1) : y - 1;
}

// ------------------------------------------------------------------------------------------
// PERIODIC PULL, vector form.  Replaces the "pull periodic" dispatch (pull-push.cpp:268-288) and
// its device function pull_periodic (:5-73).  Grid: x = 16-byte chunks of a row, y/z = rows.
// Requires nx % V == 0, nx >= 2V.
//
// All nine loads are issued back to back with no branch between them.  The shifted loads of the
// first and last lane of a row reach one element outside the row; that element is still inside
// the array (channels 1,5,8 are never plane 0, channels 3,6,7 never plane 8), its value is
// replaced afterwards by the periodic neighbour, and only those two lanes take that branch.
template <typename T, int V, int BLOCK, int ARITH = kArithUnrolled>
__global__ void __launch_bounds__(BLOCK)
k_pull_periodic_vec(const T *__restrict__ f, T *__restrict__ buf, const Geom g, const T om)
{
    uint32_t chunk, y;
    block_coords(g, chunk, y);
    const uint32_t x0 = (chunk * BLOCK + threadIdx.x) * V;
    if (x0 >= g.nx || y >= g.y_hi) return;
    const size_t plane = (size_t)g.ny * g.nx;
    const uint32_t yu = row_up(g, y), yd = row_down(g, y);
    const T *r7 = rowp(f, plane, 7, yu, g.nx), *r4 = rowp(f, plane, 4, yu, g.nx), *r8 = rowp(f, plane, 8, yu, g.nx);
    const T *r3 = rowp(f, plane, 3, y, g.nx), *r0 = rowp(f, plane, 0, y, g.nx), *r1 = rowp(f, plane, 1, y, g.nx);
    const T *r6 = rowp(f, plane, 6, yd, g.nx), *r2 = rowp(f, plane, 2, yd, g.nx), *r5 = rowp(f, plane, 5, yd, g.nx);
    const ptrdiff_t xl = (ptrdiff_t)x0 - 1, xr = (ptrdiff_t)x0 + 1;
    T c0[V], c1[V], c2[V], c3[V], c4[V], c5[V], c6[V], c7[V], c8[V];
    // gather: pull-push.cpp:13-21
    load_u<T, V>(c7, r7 + xr);
    load_a<T, V>(c4, r4 + x0);
    load_u<T, V>(c8, r8 + xl);
    load_u<T, V>(c3, r3 + xr);
    load_a<T, V>(c0, r0 + x0);
    load_u<T, V>(c1, r1 + xl);
    load_u<T, V>(c6, r6 + xr);
    load_a<T, V>(c2, r2 + x0);
    load_u<T, V>(c5, r5 + xl);
    if (x0 == 0) {  // periodic wrap, pull-push.cpp:278
        c8[0] = r8[g.nx - 1];
        c1[0] = r1[g.nx - 1];
        c5[0] = r5[g.nx - 1];
    }
    if (x0 + V == g.nx) {  // pull-push.cpp:277
        c7[V - 1] = r7[0];
        c3[V - 1] = r3[0];
        c6[V - 1] = r6[0];
    }
    T o[9][V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        T r[9];
        collide<T, ARITH>(c0[i], c1[i], c2[i], c3[i], c4[i], c5[i], c6[i], c7[i], c8[i], om, r);
#pragma unroll
        for (int q = 0; q < 9; ++q) o[q][i] = r[q];
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) store_a<T, V>(rowp(buf, plane, q, y, g.nx) + x0, o[q]);
}

// PERIODIC PUSH, vector form.  Replaces "push periodic" (pull-push.cpp:246-266) and push_periodic
// (:124-186).  A plain push writes the six x-moving populations shifted by one element, i.e. with
// 16-byte stores that straddle cache lines and leave the first and last line of every 1 KiB chunk
// half written by two different waves; measured, those partial lines cost 9 % against the pull
// sweep although the byte counts are equal.  Here every store is a whole aligned vector instead:
//   * each lane relaxes its V nodes and hands the value that crosses into the neighbouring lane's
//     vector over with a wavefront shuffle;
//   * the value that crosses into the wave from outside comes from a halo node: lane 0 also
//     relaxes the node left of the chunk, the last lane the node right of it (periodic in x), so
//     a wave writes exactly the lines it owns.  The two halo nodes cost 9 dword loads by two lanes
//     and one extra pass of the relaxation arithmetic.
template <typename T, int V, int BLOCK, int ARITH = kArithUnrolled>
__global__ void __launch_bounds__(BLOCK)
k_push_periodic_vec(const T *__restrict__ f, T *__restrict__ buf, const Geom g, const T om)
{
    uint32_t chunk, y;
    block_coords(g, chunk, y);
    const uint32_t x0 = (chunk * BLOCK + threadIdx.x) * V;
    if (x0 >= g.nx || y >= g.y_hi) return;
    const size_t plane = (size_t)g.ny * g.nx;
    const uint32_t yu = row_up(g, y), yd = row_down(g, y);
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const bool first = (lane == 0);                              // left end of this wave's chunk
    const bool last = (lane == kWave - 1) || (x0 + V == g.nx);   // right end (or end of the row)
    const T *rq[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) rq[q] = rowp(f, plane, q, y, g.nx);
    T c[9][V];
#pragma unroll
    for (int q = 0; q < 9; ++q) load_a<T, V>(c[q], rq[q] + x0);
    // halo node of this lane (wrap by comparison, pull-push.cpp:255-256); a wave with a single
    // active lane needs both and takes the right-hand one in a second pass
    const uint32_t hx_left = (x0 == 0) ? g.nx - 1 : x0 - 1;
    const uint32_t hx_right = (x0 + V == g.nx) ? 0u : x0 + V;
    T h[9], ho[9], ho2[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) { h[q] = T(0); ho[q] = T(0); ho2[q] = T(0); }
    if (first || last) {
        const uint32_t hx = first ? hx_left : hx_right;
#pragma unroll
        for (int q = 0; q < 9; ++q) h[q] = rq[q][hx];
    }
    T o[9][V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        T r[9];
        collide<T, ARITH>(c[0][i], c[1][i], c[2][i], c[3][i], c[4][i], c[5][i], c[6][i], c[7][i],
                   c[8][i], om, r);
#pragma unroll
        for (int q = 0; q < 9; ++q) o[q][i] = r[q];
    }
    if (first || last) collide<T, ARITH>(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], om, ho);
    if (first && last) {
        T h2[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) h2[q] = rq[q][hx_right];
        collide<T, ARITH>(h2[0], h2[1], h2[2], h2[3], h2[4], h2[5], h2[6], h2[7], h2[8], om, ho2);
    }
    // destination vectors at [x0, x0+V): +x movers take their first element from the lane to the
    // left (or the left halo), -x movers their last element from the lane to the right (or the
    // right halo).  pull-push.cpp:175-185 names the targets.
    auto from_left = [&](int q, T (&d)[V]) {
        const T nb = __shfl_up(o[q][V - 1], 1, kWave);
        d[0] = first ? ho[q] : nb;
#pragma unroll
        for (int i = 1; i < V; ++i) d[i] = o[q][i - 1];
    };
    auto from_right = [&](int q, T (&d)[V]) {
        const T nb = __shfl_down(o[q][0], 1, kWave);
        d[V - 1] = last ? (first ? ho2[q] : ho[q]) : nb;
#pragma unroll
        for (int i = 0; i < V - 1; ++i) d[i] = o[q][i + 1];
    };
    T d[V];
    from_right(7, d); store_a<T, V>(rowp(buf, plane, 7, yd, g.nx) + x0, d);
    store_a<T, V>(rowp(buf, plane, 4, yd, g.nx) + x0, o[4]);
    from_left(8, d);  store_a<T, V>(rowp(buf, plane, 8, yd, g.nx) + x0, d);
    from_right(3, d); store_a<T, V>(rowp(buf, plane, 3, y, g.nx) + x0, d);
    store_a<T, V>(rowp(buf, plane, 0, y, g.nx) + x0, o[0]);
    from_left(1, d);  store_a<T, V>(rowp(buf, plane, 1, y, g.nx) + x0, d);
    from_right(6, d); store_a<T, V>(rowp(buf, plane, 6, yu, g.nx) + x0, d);
    store_a<T, V>(rowp(buf, plane, 2, yu, g.nx) + x0, o[2]);
    from_left(5, d);  store_a<T, V>(rowp(buf, plane, 5, yu, g.nx) + x0, d);
}

// ------------------------------------------------------------------------------------------
// Single-node forms (any nx, explicit neighbour indices).  Used by the scalar kernels, by the
// halo'd inner/outer forms and by the wall lanes of the lid kernels.
template <typename T, int ARITH = kArithUnrolled>
__device__ __forceinline__ void pull_node(const T *__restrict__ f, T *__restrict__ buf, size_t plane,
                                          uint32_t nx, uint32_t x, uint32_t y, uint32_t xl,
                                          uint32_t xr, uint32_t yd, uint32_t yu, T om)
{
    const T f_7 = rowp(f, plane, 7, yu, nx)[xr];
    const T f_4 = rowp(f, plane, 4, yu, nx)[x];
    const T f_8 = rowp(f, plane, 8, yu, nx)[xl];
    const T f_3 = rowp(f, plane, 3, y, nx)[xr];
    const T f_0 = rowp(f, plane, 0, y, nx)[x];
    const T f_1 = rowp(f, plane, 1, y, nx)[xl];
    const T f_6 = rowp(f, plane, 6, yd, nx)[xr];
    const T f_2 = rowp(f, plane, 2, yd, nx)[x];
    const T f_5 = rowp(f, plane, 5, yd, nx)[xl];
    T o[9];
    collide<T, ARITH>(f_0, f_1, f_2, f_3, f_4, f_5, f_6, f_7, f_8, om, o);
#pragma unroll
    for (int q = 0; q < 9; ++q) rowp(buf, plane, q, y, nx)[x] = o[q];
}

template <typename T, int ARITH = kArithUnrolled>
__device__ __forceinline__ void push_node(const T *__restrict__ f, T *__restrict__ buf, size_t plane,
                                          uint32_t nx, uint32_t x, uint32_t y, uint32_t xl,
                                          uint32_t xr, uint32_t yd, uint32_t yu, T om)
{
    T c[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) c[q] = rowp(f, plane, q, y, nx)[x];
    T o[9];
    collide<T, ARITH>(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], om, o);
    rowp(buf, plane, 7, yd, nx)[xl] = o[7];
    rowp(buf, plane, 4, yd, nx)[x] = o[4];
    rowp(buf, plane, 8, yd, nx)[xr] = o[8];
    rowp(buf, plane, 3, y, nx)[xl] = o[3];
    rowp(buf, plane, 0, y, nx)[x] = o[0];
    rowp(buf, plane, 1, y, nx)[xr] = o[1];
    rowp(buf, plane, 6, yu, nx)[xl] = o[6];
    rowp(buf, plane, 2, yu, nx)[x] = o[2];
    rowp(buf, plane, 5, yu, nx)[xr] = o[5];
}

// Streaming part of one node of push_lid_driven: pull-push.cpp:496-528.  l/r/t/b: the node touches
// the left / right / top (lid) / bottom wall.  df = rho_eq*u_lid/6 (:518), applied only where t.
// Targets whose x lies outside [x_min, x_max] are skipped (the caller's neighbour lane writes them).
template <typename T>
__device__ __forceinline__ void lid_push_scatter(T *__restrict__ buf, size_t plane, uint32_t nx,
                                                 uint32_t x, uint32_t y, bool l, bool r, bool t, bool b,
                                                 const T (&o)[9], T df_lid, uint32_t x_min, uint32_t x_max)
{
    const bool lb = l || b, lt = l || t, rb = r || b, rt = r || t;
    const uint32_t xl = l ? x : x - 1, xr = r ? x : x + 1;
    const uint32_t yu = t ? y : y + 1, yd = b ? y : y - 1;
    const T df = t ? df_lid : T(0.);
    auto put = [&](int q, uint32_t yy, uint32_t xx, T v) {
        if (xx >= x_min && xx <= x_max) rowp(buf, plane, q, yy, nx)[xx] = v;
    };
    put(lb ? 5 : 7, lb ? y : yd, lb ? x : xl, o[7]);
    put(l ? 1 : 3, y, xl, o[3]);
    put(lt ? 8 : 6, lt ? y : yu, lt ? x : xl, o[6] + df);
    put(b ? 2 : 4, yd, x, o[4]);
    put(0, y, x, o[0]);
    put(t ? 4 : 2, yu, x, o[2]);
    put(rb ? 6 : 8, rb ? y : yd, rb ? x : xr, o[8]);
    put(r ? 3 : 1, y, xr, o[1]);
    put(rt ? 7 : 5, rt ? y : yu, rt ? x : xr, o[5] - df);
}

// One node of push_lid_driven: pull-push.cpp:436-528.
template <typename T, int ARITH = kArithUnrolled>
__device__ __forceinline__ void lid_push_node(const T *__restrict__ f, T *__restrict__ buf,
                                              size_t plane, uint32_t nx, uint32_t x, uint32_t y,
                                              bool l, bool r, bool t, bool b, T om, T df_lid)
{
    T c[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) c[q] = rowp(f, plane, q, y, nx)[x];
    T o[9];
    collide<T, ARITH>(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], om, o);
    lid_push_scatter<T>(buf, plane, nx, x, y, l, r, t, b, o, df_lid, 0u, nx - 1);
}

// One node of the lid-driven cavity in pull form: the store map of pull-push.cpp:520-528 read from
// the destination side (no reference counterpart).  A population whose upstream node lies behind
// a wall is the node's own opposite population; the two that bounce off the lid carry -+df.
template <typename T, int ARITH = kArithUnrolled>
__device__ __forceinline__ void lid_pull_node(const T *__restrict__ f, T *__restrict__ buf,
                                              size_t plane, uint32_t nx, uint32_t x, uint32_t y,
                                              bool l, bool r, bool t, bool b, T om, T df_lid)
{
    const T df = t ? df_lid : T(0.);
    const T f_0 = rowp(f, plane, 0, y, nx)[x];
    const T f_1 = l ? rowp(f, plane, 3, y, nx)[x] : rowp(f, plane, 1, y, nx)[x - 1];
    const T f_3 = r ? rowp(f, plane, 1, y, nx)[x] : rowp(f, plane, 3, y, nx)[x + 1];
    const T f_2 = b ? rowp(f, plane, 4, y, nx)[x] : rowp(f, plane, 2, y - 1, nx)[x];
    const T f_4 = t ? rowp(f, plane, 2, y, nx)[x] : rowp(f, plane, 4, y + 1, nx)[x];
    const T f_5 = (l || b) ? rowp(f, plane, 7, y, nx)[x] : rowp(f, plane, 5, y - 1, nx)[x - 1];
    const T f_6 = (r || b) ? rowp(f, plane, 8, y, nx)[x] : rowp(f, plane, 6, y - 1, nx)[x + 1];
    const T f_7 = (r || t) ? (rowp(f, plane, 5, y, nx)[x] - df) : rowp(f, plane, 7, y + 1, nx)[x + 1];
    const T f_8 = (l || t) ? (rowp(f, plane, 6, y, nx)[x] + df) : rowp(f, plane, 8, y + 1, nx)[x - 1];
    T o[9];
    collide<T, ARITH>(f_0, f_1, f_2, f_3, f_4, f_5, f_6, f_7, f_8, om, o);
#pragma unroll
    for (int q = 0; q < 9; ++q) rowp(buf, plane, q, y, nx)[x] = o[q];
}

// ------------------------------------------------------------------------------------------
// LID-DRIVEN sweeps, vector form.  Rows that touch the bottom wall or the lid, and the two lanes
// of a row whose 16 bytes touch a side wall, stream node by node with the reference's target map;
// every other lane runs the periodic sweeps' vector code (no wrap can occur there).
//
// PULL: shifted vector loads, aligned stores (as k_pull_periodic_vec).
// PUSH: aligned loads and whole aligned stores built with wavefront shuffles and one halo node at
// each end of the wave's chunk (as k_push_periodic_vec).  Every lane writes exactly the x range it
// owns: the wall lanes' node-wise scatter skips the one target that belongs to the neighbouring
// lane's vector and adds the one element per channel that arrives from that neighbour.
template <typename T, int V, int BLOCK, bool PULL, int ARITH = kArithUnrolled>
__global__ void __launch_bounds__(BLOCK)
k_lid_vec(const T *__restrict__ f, T *__restrict__ buf, const Geom g, const T om, const T df_lid)
{
    uint32_t chunk, y;
    block_coords(g, chunk, y);
    const uint32_t x0 = (chunk * BLOCK + threadIdx.x) * V;
    if (x0 >= g.nx || y >= g.y_hi) return;
    const size_t plane = (size_t)g.ny * g.nx;
    const uint64_t yg = (uint64_t)((int64_t)y + g.y_goff);
    const bool t = (yg == g.ny_glob - 1), b = (yg == 0);
    const bool wall_lane = (x0 == 0) || (x0 + V == g.nx);
    if (PULL) {
        if (t || b) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const uint32_t x = x0 + i;
                lid_pull_node<T, (ARITH == kArithRolled ? kArithUnrolled : ARITH)>(f, buf, plane, g.nx, x, y, x == 0, x == g.nx - 1, t, b, om, df_lid);
            }
            return;
        }
        // rows away from the bottom wall and the lid: the periodic pull code; the shifted loads of
        // the two side-wall lanes reach one element outside the row (still inside the array, see
        // k_pull_periodic_vec) and that element is replaced by the node's own opposite population
        // (half-way bounce-back; the lid term is zero in these rows but is added as the reference
        // adds it, pull-push.cpp:522,528)
        const uint32_t yu = y + 1, yd = y - 1;
        const T *r7 = rowp(f, plane, 7, yu, g.nx), *r4 = rowp(f, plane, 4, yu, g.nx), *r8 = rowp(f, plane, 8, yu, g.nx);
        const T *r3 = rowp(f, plane, 3, y, g.nx), *r0 = rowp(f, plane, 0, y, g.nx), *r1 = rowp(f, plane, 1, y, g.nx);
        const T *r6 = rowp(f, plane, 6, yd, g.nx), *r2 = rowp(f, plane, 2, yd, g.nx), *r5 = rowp(f, plane, 5, yd, g.nx);
        const ptrdiff_t xl = (ptrdiff_t)x0 - 1, xr = (ptrdiff_t)x0 + 1;
        T c[9][V];
        load_u<T, V>(c[7], r7 + xr);
        load_a<T, V>(c[4], r4 + x0);
        load_u<T, V>(c[8], r8 + xl);
        load_u<T, V>(c[3], r3 + xr);
        load_a<T, V>(c[0], r0 + x0);
        load_u<T, V>(c[1], r1 + xl);
        load_u<T, V>(c[6], r6 + xr);
        load_a<T, V>(c[2], r2 + x0);
        load_u<T, V>(c[5], r5 + xl);
        const T df0 = T(0.);
        if (x0 == 0) {              // left wall: 1 <- 3, 5 <- 7, 8 <- 6 of the node itself
            c[1][0] = rowp(f, plane, 3, y, g.nx)[0];
            c[5][0] = rowp(f, plane, 7, y, g.nx)[0];
            c[8][0] = rowp(f, plane, 6, y, g.nx)[0] + df0;
        }
        if (x0 + V == g.nx) {       // right wall: 3 <- 1, 6 <- 8, 7 <- 5
            c[3][V - 1] = rowp(f, plane, 1, y, g.nx)[g.nx - 1];
            c[6][V - 1] = rowp(f, plane, 8, y, g.nx)[g.nx - 1];
            c[7][V - 1] = rowp(f, plane, 5, y, g.nx)[g.nx - 1] - df0;
        }
        T o[9][V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
            T r[9];
            collide<T, ARITH>(c[0][i], c[1][i], c[2][i], c[3][i], c[4][i], c[5][i], c[6][i], c[7][i],
                       c[8][i], om, r);
#pragma unroll
            for (int q = 0; q < 9; ++q) o[q][i] = r[q];
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) store_a<T, V>(rowp(buf, plane, q, y, g.nx) + x0, o[q]);
        return;
    }

    // ---- push ----
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const bool first = (lane == 0) && (x0 != 0);                       // needs the node left of the chunk
    const bool last = (lane == kWave - 1) && (x0 + V != g.nx);         // needs the node right of it
    const T *rq[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) rq[q] = rowp(f, plane, q, y, g.nx);
    T c[9][V];
#pragma unroll
    for (int q = 0; q < 9; ++q) load_a<T, V>(c[q], rq[q] + x0);
    T h[9], ho[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) { h[q] = T(0); ho[q] = T(0); }
    if (first || last) {
        const uint32_t hx = first ? x0 - 1 : x0 + V;
#pragma unroll
        for (int q = 0; q < 9; ++q) h[q] = rq[q][hx];
    }
    T o[9][V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        T r[9];
        collide<T, ARITH>(c[0][i], c[1][i], c[2][i], c[3][i], c[4][i], c[5][i], c[6][i], c[7][i],
                   c[8][i], om, r);
#pragma unroll
        for (int q = 0; q < 9; ++q) o[q][i] = r[q];
    }
    if (first || last) collide<T, ARITH>(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], om, ho);
    // shuffles are executed by every lane of the wave, wall lanes included (their neighbours read
    // from them)
    T d[6][V];
    const int qs_left[3] = {8, 1, 5};    // +x movers: first element comes from the left
    const int qs_right[3] = {7, 3, 6};   // -x movers: last element comes from the right
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int q = qs_left[k];
        const T nb = __shfl_up(o[q][V - 1], 1, kWave);
        d[k][0] = first ? ho[q] : nb;
#pragma unroll
        for (int i = 1; i < V; ++i) d[k][i] = o[q][i - 1];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int q = qs_right[k];
        const T nb = __shfl_down(o[q][0], 1, kWave);
        d[3 + k][V - 1] = last ? ho[q] : nb;
#pragma unroll
        for (int i = 0; i < V - 1; ++i) d[3 + k][i] = o[q][i + 1];
    }
    if (t || b || wall_lane) {
        // node-wise streaming with the wall map.  In wall rows every lane does this, so all targets
        // are written; in other rows only the side-wall lanes do, and the target that crosses into
        // the neighbouring lane's vector is left to that lane.
        const bool whole_row = t || b;
        const uint32_t x_min = whole_row ? 0u : x0, x_max = whole_row ? g.nx - 1 : x0 + V - 1;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const uint32_t x = x0 + i;
            T oi[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) oi[q] = o[q][i];
            lid_push_scatter<T>(buf, plane, g.nx, x, y, x == 0, x == g.nx - 1, t, b, oi, df_lid, x_min, x_max);
        }
        if (!whole_row) {
            // ... and the one element per x-moving channel that arrives in this lane's range from
            // the neighbouring lane's nodes
            const uint32_t yu = y + 1, yd = y - 1;
            if (x0 != 0) {            // right-wall lane: +x movers from the lane to the left
                rowp(buf, plane, 8, yd, g.nx)[x0] = d[0][0];
                rowp(buf, plane, 1, y, g.nx)[x0] = d[1][0];
                rowp(buf, plane, 5, yu, g.nx)[x0] = d[2][0];
            }
            if (x0 + V != g.nx) {     // left-wall lane: -x movers from the lane to the right
                rowp(buf, plane, 7, yd, g.nx)[x0 + V - 1] = d[3][V - 1];
                rowp(buf, plane, 3, y, g.nx)[x0 + V - 1] = d[4][V - 1];
                rowp(buf, plane, 6, yu, g.nx)[x0 + V - 1] = d[5][V - 1];
            }
        }
        return;
    }
    // the lid term is zero in these rows; it is still added as the reference adds it
    // (pull-push.cpp:522,528), which only matters for the sign of an exact zero
#pragma unroll
    for (int i = 0; i < V; ++i) {
        d[5][i] = d[5][i] + T(0.);
        d[2][i] = d[2][i] - T(0.);
    }
    const uint32_t yu = y + 1, yd = y - 1;
    store_a<T, V>(rowp(buf, plane, 7, yd, g.nx) + x0, d[3]);
    store_a<T, V>(rowp(buf, plane, 4, yd, g.nx) + x0, o[4]);
    store_a<T, V>(rowp(buf, plane, 8, yd, g.nx) + x0, d[0]);
    store_a<T, V>(rowp(buf, plane, 3, y, g.nx) + x0, d[4]);
    store_a<T, V>(rowp(buf, plane, 0, y, g.nx) + x0, o[0]);
    store_a<T, V>(rowp(buf, plane, 1, y, g.nx) + x0, d[1]);
    store_a<T, V>(rowp(buf, plane, 6, yu, g.nx) + x0, d[5]);
    store_a<T, V>(rowp(buf, plane, 2, yu, g.nx) + x0, o[2]);
    store_a<T, V>(rowp(buf, plane, 5, yu, g.nx) + x0, d[2]);
}

// ------------------------------------------------------------------------------------------
// Vector sweeps for rows that cannot be accessed with naturally aligned 16-byte vectors (NX not a
// multiple of V, e.g. the 22627 x 22627 fp64 lattice of the reference's benchmark.py:12): every
// access is an element-aligned 128-bit vector, shifted where the stencil asks for it.  The first
// lane of a row, the partial last lane and (lid scenes) the wall rows stream node by node.
// MODE as ScalarMode below.
enum ScalarMode { kPullPeriodic = 0, kPushPeriodic = 1, kPullLid = 2, kPushLid = 3 };

template <typename T, int V, int MODE, int BLOCK, int ARITH = kArithUnrolled>
__global__ void __launch_bounds__(BLOCK)
k_sweep_vec_any(const T *__restrict__ f, T *__restrict__ buf, const Geom g, const T om, const T df_lid)
{
    uint32_t chunk, y;
    block_coords(g, chunk, y);
    const uint32_t x0 = (chunk * BLOCK + threadIdx.x) * V;
    if (x0 >= g.nx || y >= g.y_hi) return;
    const size_t plane = (size_t)g.ny * g.nx;
    constexpr bool LID = (MODE == kPullLid || MODE == kPushLid);
    constexpr bool PULL = (MODE == kPullPeriodic || MODE == kPullLid);
    const uint64_t yg = (uint64_t)((int64_t)y + g.y_goff);
    const bool t = LID && (yg == g.ny_glob - 1), b = LID && (yg == 0);
    const uint32_t rem = g.nx - x0;
    if (x0 == 0 || rem <= (uint32_t)V || t || b) {
        // node by node, with the wrap (pull-push.cpp:255-258, :277-280) or the wall map
        const uint32_t n = rem < (uint32_t)V ? rem : (uint32_t)V;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t x = x0 + i;
            if (LID) {
                const bool l = (x == 0), r = (x == g.nx - 1);
                if (PULL) lid_pull_node<T, (ARITH == kArithRolled ? kArithUnrolled : ARITH)>(f, buf, plane, g.nx, x, y, l, r, t, b, om, df_lid);
                else      lid_push_node<T, (ARITH == kArithRolled ? kArithUnrolled : ARITH)>(f, buf, plane, g.nx, x, y, l, r, t, b, om, df_lid);
            } else {
                const uint32_t xr = (x + 1 == g.nx) ? 0u : x + 1;
                const uint32_t xl = (x == 0) ? g.nx - 1 : x - 1;
                const uint32_t yu = row_up(g, y), yd = row_down(g, y);
                if (PULL) pull_node<T, ARITH>(f, buf, plane, g.nx, x, y, xl, xr, yd, yu, om);
                else      push_node<T, ARITH>(f, buf, plane, g.nx, x, y, xl, xr, yd, yu, om);
            }
        }
        return;
    }
    // here 0 < x0 and x0 + V < nx: no wrap, no wall inside the reach of this lane
    const uint32_t yu = LID ? y + 1 : row_up(g, y), yd = LID ? y - 1 : row_down(g, y);
    T c[9][V];
    if (PULL) {
        load_u<T, V>(c[7], rowp(f, plane, 7, yu, g.nx) + (x0 + 1));
        load_u<T, V>(c[4], rowp(f, plane, 4, yu, g.nx) + x0);
        load_u<T, V>(c[8], rowp(f, plane, 8, yu, g.nx) + (x0 - 1));
        load_u<T, V>(c[3], rowp(f, plane, 3, y, g.nx) + (x0 + 1));
        load_u<T, V>(c[0], rowp(f, plane, 0, y, g.nx) + x0);
        load_u<T, V>(c[1], rowp(f, plane, 1, y, g.nx) + (x0 - 1));
        load_u<T, V>(c[6], rowp(f, plane, 6, yd, g.nx) + (x0 + 1));
        load_u<T, V>(c[2], rowp(f, plane, 2, yd, g.nx) + x0);
        load_u<T, V>(c[5], rowp(f, plane, 5, yd, g.nx) + (x0 - 1));
    } else {
#pragma unroll
        for (int q = 0; q < 9; ++q) load_u<T, V>(c[q], rowp(f, plane, q, y, g.nx) + x0);
    }
    T o[9][V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        T r[9];
        collide<T, ARITH>(c[0][i], c[1][i], c[2][i], c[3][i], c[4][i], c[5][i], c[6][i], c[7][i],
                   c[8][i], om, r);
#pragma unroll
        for (int q = 0; q < 9; ++q) o[q][i] = r[q];
    }
    if (PULL) {
#pragma unroll
        for (int q = 0; q < 9; ++q) store_u<T, V>(rowp(buf, plane, q, y, g.nx) + x0, o[q]);
    } else {
        if (LID) {  // zero lid term, added as the reference adds it (pull-push.cpp:522,528)
#pragma unroll
            for (int i = 0; i < V; ++i) { o[6][i] = o[6][i] + T(0.); o[5][i] = o[5][i] - T(0.); }
        }
        store_u<T, V>(rowp(buf, plane, 7, yd, g.nx) + (x0 - 1), o[7]);
        store_u<T, V>(rowp(buf, plane, 4, yd, g.nx) + x0, o[4]);
        store_u<T, V>(rowp(buf, plane, 8, yd, g.nx) + (x0 + 1), o[8]);
        store_u<T, V>(rowp(buf, plane, 3, y, g.nx) + (x0 - 1), o[3]);
        store_u<T, V>(rowp(buf, plane, 0, y, g.nx) + x0, o[0]);
        store_u<T, V>(rowp(buf, plane, 1, y, g.nx) + (x0 + 1), o[1]);
        store_u<T, V>(rowp(buf, plane, 6, yu, g.nx) + (x0 - 1), o[6]);
        store_u<T, V>(rowp(buf, plane, 2, yu, g.nx) + x0, o[2]);
        store_u<T, V>(rowp(buf, plane, 5, yu, g.nx) + (x0 + 1), o[5]);
    }
}

// ------------------------------------------------------------------------------------------
// Scalar kernels: one node per lane over the rectangle [x_lo,x_hi) x [y_lo,y_hi).  Any nx.

// ARITH: spelling of the relaxation (kArithRolled for the periodic modes only; the lid modes have
// the unrolled spelling and its contraction).
template <typename T, int MODE, int BLOCK, int ARITH = kArithUnrolled>
__global__ void __launch_bounds__(BLOCK)
k_sweep_scalar(const T *__restrict__ f, T *__restrict__ buf, const Geom g, uint32_t x_lo,
               uint32_t x_hi, int wrap_x, const T om, const T df_lid)
{
    const uint32_t x = x_lo + blockIdx.x * BLOCK + threadIdx.x;
    if (x >= x_hi) return;
    const size_t plane = (size_t)g.ny * g.nx;
    const uint32_t y = block_row(g);
    if (y >= g.y_hi) return;
    {
        if (MODE == kPullPeriodic || MODE == kPushPeriodic) {
            // wrap by comparison: pull-push.cpp:255-258 / :277-280
            const uint32_t xr = wrap_x ? ((x + 1 == g.nx) ? 0u : x + 1) : x + 1;
            const uint32_t xl = wrap_x ? ((x == 0) ? g.nx - 1 : x - 1) : x - 1;
            const uint32_t yu = row_up(g, y), yd = row_down(g, y);
            if (MODE == kPullPeriodic) pull_node<T, ARITH>(f, buf, plane, g.nx, x, y, xl, xr, yd, yu, om);
            else                       push_node<T, ARITH>(f, buf, plane, g.nx, x, y, xl, xr, yd, yu, om);
        } else {
            const uint64_t yg = (uint64_t)((int64_t)y + g.y_goff);
            const bool t = (yg == g.ny_glob - 1), b = (yg == 0);
            const bool l = (x == 0), r = (x == g.nx - 1);
            constexpr int A = ARITH == kArithRolled ? kArithUnrolled : ARITH;
            if (MODE == kPullLid) lid_pull_node<T, A>(f, buf, plane, g.nx, x, y, l, r, t, b, om, df_lid);
            else                  lid_push_node<T, A>(f, buf, plane, g.nx, x, y, l, r, t, b, om, df_lid);
        }
    }
}

// Ring of a halo'd array: pull_periodic_outer / push_periodic_outer (pull-push.cpp:335-429).
// Lane i < nx-2 handles x=i+1 at y=1 and y=ny-2; lane nx-2+j, j < ny-4, handles y=j+2 at x=1 and
// x=nx-2.
template <typename T, bool PULL, int BLOCK, int ARITH = kArithUnrolled>
__global__ void __launch_bounds__(BLOCK)
k_outer_ring(const T *__restrict__ f, T *__restrict__ buf, uint32_t nx, uint32_t ny, const T om)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const size_t plane = (size_t)ny * nx;
    const uint32_t n_tb = nx - 2, n_lr = (ny >= 4) ? ny - 4 : 0;
    uint32_t xa, ya, xb, yb;
    if (i < n_tb) {
        xa = xb = i + 1; ya = 1; yb = ny - 2;
    } else if (i < n_tb + n_lr) {
        ya = yb = i - n_tb + 2; xa = 1; xb = nx - 2;
    } else {
        return;
    }
    if (PULL) {
        pull_node<T, ARITH>(f, buf, plane, nx, xa, ya, xa - 1, xa + 1, ya - 1, ya + 1, om);
        if (xb != xa || yb != ya)
            pull_node<T, ARITH>(f, buf, plane, nx, xb, yb, xb - 1, xb + 1, yb - 1, yb + 1, om);
    } else {
        push_node<T, ARITH>(f, buf, plane, nx, xa, ya, xa - 1, xa + 1, ya - 1, ya + 1, om);
        if (xb != xa || yb != ya)
            push_node<T, ARITH>(f, buf, plane, nx, xb, yb, xb - 1, xb + 1, yb - 1, yb + 1, om);
    }
}

// ------------------------------------------------------------------------------------------
// pull_periodic_inner (pull-push.cpp:291-310 over the rectangle [LX,HX) x [g.y_lo,g.y_hi) of an array
// with a halo): no node of the rectangle touches the array edge, so nothing wraps.  A lane takes V
// nodes from LX + lane * V with element-aligned 128-bit accesses; the lane that would cross HX goes
// node by node.  2.2 x the element-wise kernel at 16384^2 fp32.
template <typename T, int V, int BLOCK, int ARITH = kArithUnrolled>
__global__ void __launch_bounds__(BLOCK)
k_pull_inner_vec(const T *__restrict__ f, T *__restrict__ buf, const Geom g, const uint32_t LX, const uint32_t HX,
                 const T om)
{
    uint32_t chunk, y;
    block_coords(g, chunk, y);
    const uint32_t x0 = LX + (chunk * BLOCK + threadIdx.x) * V;
    if (x0 >= HX || y >= g.y_hi) return;
    const size_t plane = (size_t)g.ny * g.nx;
    const uint32_t yu = y + 1, yd = y - 1;
    if (x0 + V > HX) {
        for (uint32_t x = x0; x < HX; ++x) pull_node<T, ARITH>(f, buf, plane, g.nx, x, y, x - 1, x + 1, yd, yu, om);
        return;
    }
    T c[9][V];
    load_u<T, V>(c[7], rowp(f, plane, 7, yu, g.nx) + (x0 + 1));
    load_u<T, V>(c[4], rowp(f, plane, 4, yu, g.nx) + x0);
    load_u<T, V>(c[8], rowp(f, plane, 8, yu, g.nx) + (x0 - 1));
    load_u<T, V>(c[3], rowp(f, plane, 3, y, g.nx) + (x0 + 1));
    load_u<T, V>(c[0], rowp(f, plane, 0, y, g.nx) + x0);
    load_u<T, V>(c[1], rowp(f, plane, 1, y, g.nx) + (x0 - 1));
    load_u<T, V>(c[6], rowp(f, plane, 6, yd, g.nx) + (x0 + 1));
    load_u<T, V>(c[2], rowp(f, plane, 2, yd, g.nx) + x0);
    load_u<T, V>(c[5], rowp(f, plane, 5, yd, g.nx) + (x0 - 1));
    T o[9][V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        T r[9];
        collide<T, ARITH>(c[0][i], c[1][i], c[2][i], c[3][i], c[4][i], c[5][i], c[6][i], c[7][i], c[8][i], om, r);
#pragma unroll
        for (int q = 0; q < 9; ++q) o[q][i] = r[q];
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) store_u<T, V>(rowp(buf, plane, q, y, g.nx) + x0, o[q]);
}

// ------------------------------------------------------------------------------------------
// Initialisation: broadcast a per-row table along x.  table is [9][rows]; array rows
// [y_first, y_first+rows) are written.  Replaces the "initialize" / "initialize at rest"
// dispatches (init.cpp:14-33, :37-48); the table holds what f_eq (init.cpp:57-84) returns.
// AOS: f is [NY][NX][9] (global.h:40) instead of [9][NY][NX].
template <typename T, int BLOCK, bool AOS = false>
__global__ void __launch_bounds__(BLOCK)
k_fill_rows(T *__restrict__ f, uint32_t nx, uint32_t ny, uint32_t y_first, uint32_t rows,
            const T *__restrict__ table)
{
    const uint32_t x = blockIdx.x * BLOCK + threadIdx.x;
    if (x >= nx) return;
    const size_t plane = (size_t)ny * nx;
    for (uint32_t r = blockIdx.y; r < rows; r += gridDim.y) {
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            if (AOS) f[((size_t)(y_first + r) * nx + x) * 9 + q] = table[q * rows + r];
            else     rowp(f, plane, q, y_first + r, nx)[x] = table[q * rows + r];
        }
    }
}

// The same for [9][NY][NX] rows that are whole 16-byte vectors long: a lane broadcasts the row's
// value into one non-temporal 128-bit store per population, a wavefront fills 1 KiB per store
// (16384^2: fp32 6.3 TB/s against 6.1, fp64 5.85 against 5.73).
template <typename T, int V, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_fill_rows_vec(T *__restrict__ f, uint32_t nx, uint32_t ny, uint32_t y_first, uint32_t rows,
                const T *__restrict__ table)
{
    const uint32_t v = blockIdx.x * BLOCK + threadIdx.x;
    if (v >= nx / V) return;
    const size_t plane = (size_t)ny * nx;
    for (uint32_t r = blockIdx.y; r < rows; r += gridDim.y) {
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const T t = table[q * rows + r];
            T vec[V];
#pragma unroll
            for (int i = 0; i < V; ++i) vec[i] = t;
            store_a<T, V>(rowp(f, plane, q, y_first + r, nx) + (size_t)v * V, vec);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Output path.  WHAT: 0 = ux,uy planes (compute_velocities, io.cpp:136-174), 1 = |u| plane
// (io.cpp:236-238).
template <typename T, int WHAT, int BLOCK, bool AOS = false>
__global__ void __launch_bounds__(BLOCK)
k_velocity(const T *__restrict__ f, T *__restrict__ out, uint32_t nx, uint32_t ny, uint32_t y_lo,
           uint32_t y_hi)
{
    const uint32_t x = blockIdx.x * BLOCK + threadIdx.x;
    if (x >= nx) return;
    const size_t plane = (size_t)ny * nx;
    const uint32_t rows = y_hi - y_lo;
    for (uint32_t y = y_lo + blockIdx.y; y < y_hi; y += gridDim.y) {
        T c[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) c[q] = AOS ? f[((size_t)y * nx + x) * 9 + q] : rowp(f, plane, q, y, nx)[x];
        T ux, uy;
        node_velocity<T>(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], ux, uy);
        const size_t o = (size_t)(y - y_lo) * nx + x;
        if (WHAT == 0) {
            out[o] = ux;
            out[(size_t)rows * nx + o] = uy;
        } else {
            out[o] = node_speed<T>(ux, uy);
        }
    }
}

// Max |u| reduction (io.cpp:184-208).  Speeds are >= 0 or NaN, so the IEEE bit pattern of a valid
// speed orders like an unsigned integer; key = bits+1, key 0 = "no node qualified".  Lanes reduce
// with wavefront shuffles, waves through LDS, then one atomicMax per workgroup on *key.
__device__ __forceinline__ uint32_t speed_key(float u)
{
    // NaN never wins (io.cpp:205: `if (u > local_max)`); -0.0f cannot occur from sqrtf(x*x+y*y).
    return (u == u) ? (__float_as_uint(u) + 1u) : 0u;
}

template <typename T, int V, int BLOCK, bool AOS = false>
__global__ void __launch_bounds__(BLOCK)
k_max_speed(const T *__restrict__ f, uint32_t nx, uint32_t ny, uint32_t y_lo, uint32_t y_hi,
            uint32_t *__restrict__ key)
{
    const size_t plane = (size_t)ny * nx;
    uint32_t best = 0;
    const uint32_t nvec = nx / V;  // vector chunks per row (nx % V == 0 when V > 1)
    for (uint32_t y = y_lo + blockIdx.y; y < y_hi; y += gridDim.y) {
        for (uint32_t v = blockIdx.x * BLOCK + threadIdx.x; v < nvec; v += gridDim.x * BLOCK) {
            T c[9][V];
            if (AOS) {
                static_assert(!AOS || V == 1, "array-of-structures rows are read one node per lane");
#pragma unroll
                for (int q = 0; q < 9; ++q) c[q][0] = f[((size_t)y * nx + v) * 9 + q];
            } else {
#pragma unroll
                for (int q = 0; q < 9; ++q) load_a<T, V>(c[q], rowp(f, plane, q, y, nx) + (size_t)v * V);
            }
#pragma unroll
            for (int i = 0; i < V; ++i) {
                T ux, uy;
                node_velocity<T>(c[0][i], c[1][i], c[2][i], c[3][i], c[4][i], c[5][i], c[6][i],
                                 c[7][i], c[8][i], ux, uy);
                // the reference narrows to float for sqrtf and keeps the float value (global.h:30)
                const uint32_t k = speed_key(sqrtf((float)(ux * ux + uy * uy)));
                best = (k > best) ? k : best;
            }
        }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const uint32_t other = __shfl_down(best, off, kWave);
        best = (other > best) ? other : best;
    }
    __shared__ uint32_t wave_best[BLOCK / kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) wave_best[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = wave_best[0];
#pragma unroll
        for (int w = 1; w < BLOCK / kWave; ++w) m = (wave_best[w] > m) ? wave_best[w] : m;
        // the key only grows, so a value read earlier can only be too small: skipping on it is safe,
        // and most workgroups then issue no atomic (one per workgroup costs 8 % at 65536 workgroups)
        if (m != 0 && m > __hip_atomic_load(key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(key, m);
    }
}

// Decode the key into one FLT (start value -1e30 as io.cpp:184 when no node qualified).
template <typename T>
__global__ void k_max_speed_finish(const uint32_t *key, T *out)   // out may be the key's own address
{
    const uint32_t k = *key;
    *out = (k == 0) ? (T)-1e30 : (T)__uint_as_float(k - 1u);
}

// out[i] = the contracted arithmetic's reciprocal of x[i] (recip_x), for the library's self-check.
template <typename T>
__global__ void __launch_bounds__(256) k_reciprocal(const T *__restrict__ x, T *__restrict__ out, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = recip_x<T, T>(x[i]);
}

}  // namespace lbm
