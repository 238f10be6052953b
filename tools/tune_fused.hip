// Tuning harness (not product): K time steps of the periodic pull sweep fused into one pass.
// One wave marches a 1 KiB-wide strip through a segment of rows; intermediate time levels live in
// registers, x-neighbours come from wavefront shifts.  Checked bit for bit against K launches of
// the product one-step kernel.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include "lbm_device.h"
#pragma clang fp contract(off)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2);} } while (0)
using namespace lbm;

struct FGeom {
    uint32_t nx, ny, y_lo, y_hi;
    int wrap_y;
    uint32_t seg_rows;   // output rows per wave
    uint32_t nstrips, nsegs;
    int xcd_map;
};

#ifndef USE_DPP
#define USE_DPP 1
#endif

template <typename T> struct Bits;
template <> struct Bits<float> { static constexpr int n = 1; };
template <> struct Bits<double> { static constexpr int n = 2; };

// value held by lane-1 / lane+1 of the wavefront (lane 0 / 63 get an unspecified value)
template <typename T>
__device__ __forceinline__ T from_lane_below(T v)
{
#if USE_DPP
    union { T t; int w[Bits<T>::n]; } a, b;
    a.t = v;
#pragma unroll
    for (int k = 0; k < Bits<T>::n; ++k) b.w[k] = __builtin_amdgcn_update_dpp(a.w[k], a.w[k], 0x138 /*wave_shr:1*/, 0xf, 0xf, false);
    return b.t;
#else
    return __shfl_up(v, 1, 64);
#endif
}
template <typename T>
__device__ __forceinline__ T from_lane_above(T v)
{
#if USE_DPP
    union { T t; int w[Bits<T>::n]; } a, b;
    a.t = v;
#pragma unroll
    for (int k = 0; k < Bits<T>::n; ++k) b.w[k] = __builtin_amdgcn_update_dpp(a.w[k], a.w[k], 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
    return b.t;
#else
    return __shfl_down(v, 1, 64);
#endif
}

// gather + relax one row held as nine aligned vectors (already the right rows per channel)
template <typename T, int V>
__device__ __forceinline__ void stage(const T (&s)[9][V], const T om, T (&o)[9][V])
{
    T c[9][V];
#pragma unroll
    for (int i = 0; i < V; ++i) { c[0][i] = s[0][i]; c[2][i] = s[2][i]; c[4][i] = s[4][i]; }
    // channels moving in +x arrive from x-1
#pragma unroll
    for (int q : {1, 5, 8}) {
        c[q][0] = from_lane_below(s[q][V - 1]);
#pragma unroll
        for (int i = 1; i < V; ++i) c[q][i] = s[q][i - 1];
    }
#pragma unroll
    for (int q : {3, 6, 7}) {
        c[q][V - 1] = from_lane_above(s[q][0]);
#pragma unroll
        for (int i = 0; i < V - 1; ++i) c[q][i] = s[q][i + 1];
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
        T r[9];
        collide<T>(c[0][i], c[1][i], c[2][i], c[3][i], c[4][i], c[5][i], c[6][i], c[7][i], c[8][i], om, r);
#pragma unroll
        for (int q = 0; q < 9; ++q) o[q][i] = r[q];
    }
}

template <typename T, int V, int K, int H, bool NT, int PF>
__global__ void __launch_bounds__(64)
k_pull_fused(const T *__restrict__ f, T *__restrict__ buf, const FGeom g, const T om)
{
    extern __shared__ char lds[];
    if (g.nx == 0xffffffffu) lds[threadIdx.x] = 1;
    static_assert(H * V >= K, "halo too small");   // H = halo vectors on either side of a strip
    constexpr int W = 64 - 2 * H;               // output vectors per strip
    uint32_t strip, seg;
    {
        const uint32_t b = blockIdx.x;
        if (g.xcd_map) {
            const uint32_t xcd = b & 7u, slot = b >> 3;
            seg = (slot / g.nstrips) * 8 + xcd;
            strip = slot % g.nstrips;
        } else {
            strip = b % g.nstrips;
            seg = b / g.nstrips;
        }
    }
    if (seg >= g.nsegs) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t nxv = g.nx / V;
    // vector handled by this lane, periodic in x
    int64_t vx = (int64_t)strip * W - H + lane;
    vx %= (int64_t)nxv; if (vx < 0) vx += nxv;
    const uint32_t x0 = (uint32_t)vx * V;
    const uint32_t wcount = min((uint32_t)W, nxv - strip * W);
    const bool store_lane = lane >= (uint32_t)H && lane < H + wcount;
    const uint32_t ys = g.y_lo + seg * g.seg_rows;
    const uint32_t ye = min(ys + g.seg_rows, g.y_hi);
    const uint32_t iters = (ye - ys) + 2 * (K - 1);
    const size_t plane = (size_t)g.ny * g.nx;

    auto wrap = [&](int64_t y) -> uint32_t {
        if (g.wrap_y) { if (y < 0) y += g.ny; if (y >= (int64_t)g.ny) y -= g.ny; }
        return (uint32_t)y;
    };
    // stage-1 input of iteration i: f rows R-1 (2,5,6), R (0,1,3), R+1 (4,7,8), R = ys-(K-1)+i
    auto load_row = [&](T (&d)[9][V], uint32_t i) {
        const int64_t R = (int64_t)ys - (K - 1) + i;
        const uint32_t yd = wrap(R - 1), y = wrap(R), yu = wrap(R + 1);
        load_a<T, V>(d[7], rowp(f, plane, 7, yu, g.nx) + x0);
        load_a<T, V>(d[4], rowp(f, plane, 4, yu, g.nx) + x0);
        load_a<T, V>(d[8], rowp(f, plane, 8, yu, g.nx) + x0);
        load_a<T, V>(d[3], rowp(f, plane, 3, y, g.nx) + x0);
        load_a<T, V>(d[0], rowp(f, plane, 0, y, g.nx) + x0);
        load_a<T, V>(d[1], rowp(f, plane, 1, y, g.nx) + x0);
        load_a<T, V>(d[6], rowp(f, plane, 6, yd, g.nx) + x0);
        load_a<T, V>(d[2], rowp(f, plane, 2, yd, g.nx) + x0);
        load_a<T, V>(d[5], rowp(f, plane, 5, yd, g.nx) + x0);
    };

    // between stage s and s+1: A = (2,5,6) of two rows back, B = (0,1,3) of one row back,
    // C = (2,5,6) of one row back
    T ringA[K > 1 ? K - 1 : 1][3][V], ringB[K > 1 ? K - 1 : 1][3][V], ringC[K > 1 ? K - 1 : 1][3][V];
#pragma unroll
    for (int s = 0; s < K - 1; ++s)
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < V; ++i) { ringA[s][k][i] = T(1); ringB[s][k][i] = T(1); ringC[s][k][i] = T(1); }

    T in[PF + 1][9][V];
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if ((uint32_t)p < iters) load_row(in[p], p);
    for (uint32_t i0 = 0; i0 < iters; i0 += PF + 1) {
#pragma unroll
        for (int j = 0; j <= PF; ++j) {
            const uint32_t i = i0 + j;
            if (i < iters) {
                if (i + PF < iters) load_row(in[(j + PF) % (PF + 1)], i + PF);
                T o[9][V];
                stage<T, V>(in[j], om, o);
#pragma unroll
                for (int s = 0; s < K - 1; ++s) {
                    // next stage's row: (2,5,6) from A, (0,1,3) from B, (4,7,8) fresh
                    T nx_[9][V];
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        nx_[2][e] = ringA[s][0][e]; nx_[5][e] = ringA[s][1][e]; nx_[6][e] = ringA[s][2][e];
                        nx_[0][e] = ringB[s][0][e]; nx_[1][e] = ringB[s][1][e]; nx_[3][e] = ringB[s][2][e];
                        nx_[4][e] = o[4][e]; nx_[7][e] = o[7][e]; nx_[8][e] = o[8][e];
                        ringA[s][0][e] = ringC[s][0][e]; ringA[s][1][e] = ringC[s][1][e]; ringA[s][2][e] = ringC[s][2][e];
                        ringB[s][0][e] = o[0][e]; ringB[s][1][e] = o[1][e]; ringB[s][2][e] = o[3][e];
                        ringC[s][0][e] = o[2][e]; ringC[s][1][e] = o[5][e]; ringC[s][2][e] = o[6][e];
                    }
                    stage<T, V>(nx_, om, o);
                }
                if (i >= 2 * (K - 1)) {
                    const uint32_t yo = ys + i - 2 * (K - 1);
                    if (store_lane) {
#pragma unroll
                        for (int q = 0; q < 9; ++q) {
                            if (NT) store_a<T, V>(rowp(buf, plane, q, yo, g.nx) + x0, o[q]);
                            else {
                                typename VecTypes<T, V>::vec_t v;
#pragma unroll
                                for (int e = 0; e < V; ++e) v[e] = o[q][e];
                                *reinterpret_cast<typename VecTypes<T, V>::vec_a *>(rowp(buf, plane, q, yo, g.nx) + x0) = v;
                            }
                        }
                    }
                }
            }
        }
    }
}

template <typename T>
__global__ void k_diff(const T *a, const T *b, size_t n, unsigned long long *cnt)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (; i < n; i += stride) {
        const unsigned char *pa = reinterpret_cast<const unsigned char *>(&a[i]);
        const unsigned char *pb = reinterpret_cast<const unsigned char *>(&b[i]);
        bool ne = false;
        for (size_t k = 0; k < sizeof(T); ++k) ne |= pa[k] != pb[k];
        if (ne) ++c;
    }
    if (c) atomicAdd(cnt, c);
}

struct Variant { std::string name; double lups; std::function<void()> launch; std::vector<float> ms; };

template <typename T, int V, int K, int H = (K + V - 1) / V, bool NT = true, int PF = 1>
void add_fused(std::vector<Variant> &vs, T *a, T *b, uint32_t nx, uint32_t ny, T om, uint32_t seg_rows, int wpc, int xcd_map)
{
    constexpr int W = 64 - 2 * H;
    FGeom g{nx, ny, 0, ny, 1, seg_rows, (nx / V + W - 1) / W, (ny + seg_rows - 1) / seg_rows, xcd_map};
    auto kern = k_pull_fused<T, V, K, H, NT, PF>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds = (wpc >= 32) ? 0 : (160 * 1024 / wpc) & ~255u;
    const uint32_t nsegs_pad = xcd_map ? ((g.nsegs + 7) / 8) * 8 : g.nsegs;
    const dim3 grid(g.nstrips * nsegs_pad);
    char nm[96];
    snprintf(nm, sizeof nm, "fusedK%d_f%d_S%u_w%d_x%d_H%d_nt%d_pf%d", K, (int)sizeof(T) * 8, seg_rows, wpc, xcd_map, H, (int)NT, PF);
    vs.push_back({nm, (double)nx * ny * K, [=] { hipLaunchKernelGGL(kern, grid, dim3(64), lds, 0, a, b, g, om); }, {}});
}

template <typename T, int V>
void product(T *a, T *b, uint32_t nx, uint32_t ny, T om)
{
    Geom g{0, nx, ny, 0, ny, 1, 0, 0};
    const dim3 grid(nx / V / 64 + ((nx / V) % 64 ? 1 : 0), ny, 1);
    const size_t lds = (160 * 1024 / 6) & ~255u;
    auto kern = k_pull_periodic_vec<T, V, 64>;
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once = true; }
    hipLaunchKernelGGL(kern, grid, dim3(64), lds, 0, a, b, g, om);
}

template <typename T, int V>
int run(uint32_t nx, uint32_t ny, int rounds, int iters, const char *which, const char *only)
{
    const size_t n = (size_t)9 * nx * ny;
    T *a, *b, *c, *d;
    CK(hipMalloc(&a, n * sizeof(T))); CK(hipMalloc(&b, n * sizeof(T)));
    CK(hipMalloc(&c, n * sizeof(T))); CK(hipMalloc(&d, n * sizeof(T)));
    {
        const uint32_t rows = 61;
        std::vector<T> h((size_t)nx * rows);
        const double w[9] = {4. / 9, 1. / 9, 1. / 9, 1. / 9, 1. / 9, 1. / 36, 1. / 36, 1. / 36, 1. / 36};
        for (int q = 0; q < 9; ++q) {
            for (size_t i = 0; i < h.size(); ++i) h[i] = (T)(w[q] * (1.0 + 0.05 * (double)((i * 7 + q * 3 + (i / nx) * 5) % 13) / 13.0));
            for (uint32_t y = 0; y < ny; y += rows) {
                const uint32_t r = std::min(rows, ny - y);
                CK(hipMemcpy(a + ((size_t)q * ny + y) * nx, h.data(), (size_t)r * nx * sizeof(T), hipMemcpyHostToDevice));
            }
        }
    }
    const T om = (T)0.5;
    unsigned long long *cnt; CK(hipMalloc(&cnt, 8));
    std::vector<Variant> vs;
    vs.push_back({"product one step", (double)nx * ny, [=] { product<T, V>(a, b, nx, ny, om); }, {}});
    auto check = [&](int K, std::function<void()> launch, const char *nm) {
        // reference: K product steps a -> ... -> result in c
        T *src = a, *dst = c, *other = d;
        for (int k = 0; k < K; ++k) { product<T, V>(src, dst, nx, ny, om); src = dst; std::swap(dst, other); }
        CK(hipMemset(b, 0xff, n * sizeof(T)));
        launch();
        CK(hipMemset(cnt, 0, 8));
        hipLaunchKernelGGL(k_diff<T>, dim3(4096), dim3(256), 0, 0, src, b, n, cnt);
        unsigned long long hc; CK(hipMemcpy(&hc, cnt, 8, hipMemcpyDeviceToHost));
        printf("check %-36s mismatches %llu of %zu\n", nm, hc, n);
        return hc;
    };
    unsigned long long bad = 0;
    auto add_checked = [&](int K, size_t first) {
        if (only) return;
        for (size_t i = first; i < vs.size(); ++i) bad += check(K, vs[i].launch, vs[i].name.c_str());
    };
    size_t first = vs.size();
    constexpr int HA = 8 / V;   // halo vectors that keep every store a whole 64-byte line... (4 for fp32->2, 2 for fp64->4)
    constexpr int H64 = 64 / (int)sizeof(T) / V / 2;   // 32 bytes of halo either side: stores cover whole 64-byte lines
    (void)HA;
    if (strchr(which, '2')) {
        for (uint32_t S : {16u, 32u, 64u, 128u}) {
            add_fused<T, V, 2, H64, true, 1>(vs, a, b, nx, ny, om, S, 4, 0);
            add_fused<T, V, 2, H64, true, 2>(vs, a, b, nx, ny, om, S, 4, 0);
            add_fused<T, V, 2, H64, true, 3>(vs, a, b, nx, ny, om, S, 4, 0);
        }
        add_fused<T, V, 2, H64, true, 2>(vs, a, b, nx, ny, om, 32, 3, 0);
        add_fused<T, V, 2, H64, true, 2>(vs, a, b, nx, ny, om, 32, 5, 0);
        add_checked(2, first);
    }
    first = vs.size();
    if (strchr(which, '3')) {
        for (uint32_t S : {16u, 32u, 64u}) {
            add_fused<T, V, 3, H64, true, 1>(vs, a, b, nx, ny, om, S, 4, 0);
            add_fused<T, V, 3, H64, true, 2>(vs, a, b, nx, ny, om, S, 4, 0);
        }
        add_checked(3, first);
    }
    first = vs.size();
    if (strchr(which, '4')) {
        for (uint32_t S : {16u, 32u, 64u}) {
            add_fused<T, V, 4, H64, true, 1>(vs, a, b, nx, ny, om, S, 4, 0);
            add_fused<T, V, 4, H64, true, 2>(vs, a, b, nx, ny, om, S, 4, 0);
        }
        add_checked(4, first);
    }
    if (only) {
        std::vector<Variant> keep;
        for (auto &v : vs) if (strstr(v.name.c_str(), only)) keep.push_back(v);
        vs.swap(keep);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    for (int r = 0; r < rounds; ++r)
        for (auto &v : vs) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) v.launch();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms / iters);
        }
    CK(hipGetLastError());
    printf("%-36s %10s %12s %12s\n", "variant", "med ms", "med MLUPS", "best MLUPS");
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        printf("%-36s %10.3f %12.0f %12.0f\n", v.name.c_str(), v.ms[v.ms.size() / 2], v.lups / v.ms[v.ms.size() / 2] / 1e3, v.lups / v.ms[0] / 1e3);
    }
    return bad ? 1 : 0;
}

int main(int argc, char **argv)
{
    uint32_t nx = 16384, ny = 16384; int rounds = 3, iters = 10;
    const char *prec = "f32", *which = "2";
    if (argc > 1) nx = atoi(argv[1]);
    if (argc > 2) ny = atoi(argv[2]);
    if (argc > 3) prec = argv[3];
    if (argc > 4) which = argv[4];
    if (argc > 5) iters = atoi(argv[5]);
    const char *only = argc > 6 ? argv[6] : nullptr;
    if (only) rounds = 1;
    if (!strcmp(prec, "f64")) return run<double, 2>(nx, ny, rounds, iters, which, only);
    return run<float, 4>(nx, ny, rounds, iters, which, only);
}
