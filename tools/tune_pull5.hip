// Pull-sweep tuning (not product): V nodes per lane x nontemporal hints x waves per CU.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#include "lbm_device.h"
#pragma clang fp contract(off)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2);} } while (0)
using namespace lbm;

template <typename T, int V, bool NT>
__device__ __forceinline__ void ldv(T (&r)[V], const T *p, bool aligned)
{
    typedef typename VecTypes<T, V>::vec_t vec_t;
    typedef typename VecTypes<T, V>::vec_a vec_a;
    typedef typename VecTypes<T, V>::vec_u vec_u;
    vec_t v;
    if (aligned) v = NT ? __builtin_nontemporal_load(reinterpret_cast<const vec_a *>(p)) : *reinterpret_cast<const vec_a *>(p);
    else         v = NT ? __builtin_nontemporal_load(reinterpret_cast<const vec_u *>(p)) : *reinterpret_cast<const vec_u *>(p);
#pragma unroll
    for (int i = 0; i < V; ++i) r[i] = v[i];
}
template <typename T, int V, bool NT>
__device__ __forceinline__ void stv(T *p, const T (&r)[V])
{
    typedef typename VecTypes<T, V>::vec_t vec_t;
    typedef typename VecTypes<T, V>::vec_a vec_a;
    vec_t v;
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = r[i];
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<vec_a *>(p)); else *reinterpret_cast<vec_a *>(p) = v;
}

// wrap handled with modular element loads for the edge lanes only
template <typename T, int V, bool NT>
__device__ __forceinline__ void ld_shift(T (&r)[V], const T *row, uint32_t x0, uint32_t nx, int shift)
{
    const bool edge = (shift < 0) ? (x0 == 0) : (x0 + V == nx);
    if (!edge) { ldv<T, V, NT>(r, row + x0 + shift, false); }
    else {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            int64_t x = (int64_t)x0 + i + shift;
            if (x < 0) x += nx;
            if (x >= (int64_t)nx) x -= nx;
            r[i] = row[x];
        }
    }
}

template <typename T, int V, int BLOCK, bool NTL, bool NTS>
__global__ void __launch_bounds__(BLOCK)
k_pull(const T *__restrict__ f, T *__restrict__ buf, const Geom g, const T om)
{
    extern __shared__ char lds[];
    if (g.nx == 0xffffffffu) lds[threadIdx.x] = 1;
    const uint32_t x0 = (blockIdx.x * BLOCK + threadIdx.x) * V;
    if (x0 >= g.nx) return;
    const size_t plane = (size_t)g.ny * g.nx;
    for (uint32_t y = g.y_lo + blockIdx.y; y < g.y_hi; y += gridDim.y) {
        const uint32_t yu = row_up(g, y), yd = row_down(g, y);
        T c0[V], c1[V], c2[V], c3[V], c4[V], c5[V], c6[V], c7[V], c8[V];
        ld_shift<T, V, NTL>(c7, rowp(f, plane, 7, yu, g.nx), x0, g.nx, +1);
        ldv<T, V, NTL>(c4, rowp(f, plane, 4, yu, g.nx) + x0, true);
        ld_shift<T, V, NTL>(c8, rowp(f, plane, 8, yu, g.nx), x0, g.nx, -1);
        ld_shift<T, V, NTL>(c3, rowp(f, plane, 3, y, g.nx), x0, g.nx, +1);
        ldv<T, V, NTL>(c0, rowp(f, plane, 0, y, g.nx) + x0, true);
        ld_shift<T, V, NTL>(c1, rowp(f, plane, 1, y, g.nx), x0, g.nx, -1);
        ld_shift<T, V, NTL>(c6, rowp(f, plane, 6, yd, g.nx), x0, g.nx, +1);
        ldv<T, V, NTL>(c2, rowp(f, plane, 2, yd, g.nx) + x0, true);
        ld_shift<T, V, NTL>(c5, rowp(f, plane, 5, yd, g.nx), x0, g.nx, -1);
        T o[9][V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
            T r[9];
            collide<T>(c0[i], c1[i], c2[i], c3[i], c4[i], c5[i], c6[i], c7[i], c8[i], om, r);
#pragma unroll
            for (int q = 0; q < 9; ++q) o[q][i] = r[q];
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) stv<T, V, NTS>(rowp(buf, plane, q, y, g.nx) + x0, o[q]);
    }
}


template <typename T, int V, int BLOCK, bool SHIFT, bool ROWS, bool NTS, int REMAP>
__global__ void __launch_bounds__(BLOCK)
k_pull_abl(const T *__restrict__ f, T *__restrict__ buf, const Geom g, const T om)
{
    extern __shared__ char lds[];
    if (REMAP != 2 && g.nx == 0xffffffffu) lds[threadIdx.x] = 1;
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    if (REMAP == 1) {
        // blocks are dealt round-robin to the 8 XCDs in linear order; give each XCD whole rows
        const uint32_t L = blockIdx.y * gridDim.x + blockIdx.x;
        const uint32_t per_group = gridDim.x * 8;
        const uint32_t grp = L / per_group, r = L % per_group;
        by = grp * 8 + (r & 7);
        bx = r >> 3;
        if (by >= gridDim.y) return;
    }
    const uint32_t x0 = (bx * BLOCK + threadIdx.x) * V;
    if (x0 >= g.nx) return;
    const size_t plane = (size_t)g.ny * g.nx;
    const uint32_t y = g.y_lo + by;
    const uint32_t yu = ROWS ? row_up(g, y) : y, yd = ROWS ? row_down(g, y) : y;
    T c0[V], c1[V], c2[V], c3[V], c4[V], c5[V], c6[V], c7[V], c8[V];
    if (SHIFT) {
        ld_shift<T, V, false>(c7, rowp(f, plane, 7, yu, g.nx), x0, g.nx, +1);
        ld_shift<T, V, false>(c8, rowp(f, plane, 8, yu, g.nx), x0, g.nx, -1);
        ld_shift<T, V, false>(c3, rowp(f, plane, 3, y, g.nx), x0, g.nx, +1);
        ld_shift<T, V, false>(c1, rowp(f, plane, 1, y, g.nx), x0, g.nx, -1);
        ld_shift<T, V, false>(c6, rowp(f, plane, 6, yd, g.nx), x0, g.nx, +1);
        ld_shift<T, V, false>(c5, rowp(f, plane, 5, yd, g.nx), x0, g.nx, -1);
    } else {
        ldv<T, V, false>(c7, rowp(f, plane, 7, yu, g.nx) + x0, true);
        ldv<T, V, false>(c8, rowp(f, plane, 8, yu, g.nx) + x0, true);
        ldv<T, V, false>(c3, rowp(f, plane, 3, y, g.nx) + x0, true);
        ldv<T, V, false>(c1, rowp(f, plane, 1, y, g.nx) + x0, true);
        ldv<T, V, false>(c6, rowp(f, plane, 6, yd, g.nx) + x0, true);
        ldv<T, V, false>(c5, rowp(f, plane, 5, yd, g.nx) + x0, true);
    }
    ldv<T, V, false>(c4, rowp(f, plane, 4, yu, g.nx) + x0, true);
    ldv<T, V, false>(c0, rowp(f, plane, 0, y, g.nx) + x0, true);
    ldv<T, V, false>(c2, rowp(f, plane, 2, yd, g.nx) + x0, true);
    T o[9][V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        T r[9];
        collide<T>(c0[i], c1[i], c2[i], c3[i], c4[i], c5[i], c6[i], c7[i], c8[i], om, r);
#pragma unroll
        for (int q = 0; q < 9; ++q) o[q][i] = r[q];
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) stv<T, V, NTS>(rowp(buf, plane, q, y, g.nx) + x0, o[q]);
}

struct Variant { std::string name; double bytes; std::function<void()> launch; std::vector<float> ms; };

template <int V, bool NTL, bool NTS>
void add(std::vector<Variant> &vs, float *a, float *b, Geom g, float om, double bytes, int wpc)
{
    auto kern = k_pull<float, V, 64, NTL, NTS>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds = (wpc >= 32) ? 0 : (160 * 1024 / wpc) & ~255u;
    char nm[96];
    snprintf(nm, sizeof nm, "pull V%d ntl%d nts%d %2d waves/CU", V, (int)NTL, (int)NTS, wpc);
    const dim3 grid((g.nx / V + 63) / 64, g.ny);
    vs.push_back({nm, bytes, [=] { hipLaunchKernelGGL(kern, grid, dim3(64), lds, 0, a, b, g, om); }, {}});
}

template <int BLOCK, bool SHIFT, bool ROWS, bool NTS, int REMAP>
void add_abl(std::vector<Variant> &vs, float *a, float *b, Geom g, float om, double bytes, int wpc)
{
    auto kern = k_pull_abl<float, 4, BLOCK, SHIFT, ROWS, NTS, REMAP>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int bpc = wpc / (BLOCK / 64);
    const size_t lds = (wpc >= 32) ? 0 : (160 * 1024 / bpc) & ~255u;
    char nm[96];
    snprintf(nm, sizeof nm, "abl b%d shift%d rows%d nts%d remap%d %2dw/CU", BLOCK, (int)SHIFT, (int)ROWS, (int)NTS, REMAP, wpc);
    const dim3 grid((g.nx / 4 + BLOCK - 1) / BLOCK, g.ny);
    vs.push_back({nm, bytes, [=] { hipLaunchKernelGGL(kern, grid, dim3(BLOCK), lds, 0, a, b, g, om); }, {}});
}

void product_variants(std::vector<Variant> &vs, float *a, float *b, Geom g, float om, double bytes)
{
    for (int wpc : {8, 32}) {
        const size_t lds = (wpc >= 32) ? 0 : (160 * 1024 / wpc) & ~511u;
        const dim3 grid(g.nx / 4 / 64, g.ny, 1);
        char nm[64]; snprintf(nm, sizeof nm, "product k_pull_periodic_vec %dw/CU", wpc);
        vs.push_back({nm, bytes, [=] { hipLaunchKernelGGL((k_pull_periodic_vec<float, 4, 64>), grid, dim3(64), lds, 0, a, b, g, om); }, {}});
    }
}

int main(int argc, char **argv)
{
    uint32_t nx = 16384, ny = 16384; int rounds = 5, iters = 20;
    if (argc > 1) nx = atoi(argv[1]);
    if (argc > 2) ny = atoi(argv[2]);
    const size_t n = (size_t)9 * nx * ny;
    float *a, *b;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    {
        std::vector<float> h((size_t)nx * 64);
        const float w[9] = {4.f / 9, 1.f / 9, 1.f / 9, 1.f / 9, 1.f / 9, 1.f / 36, 1.f / 36, 1.f / 36, 1.f / 36};
        for (int q = 0; q < 9; ++q) {
            for (size_t i = 0; i < h.size(); ++i) h[i] = w[q] * (1.0f + 0.01f * ((i * 7 + q) % 13) / 13.0f);
            for (uint32_t y = 0; y < ny; y += 64)
                CK(hipMemcpy(a + ((size_t)q * ny + y) * nx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        }
        CK(hipMemcpy(b, a, n * 4, hipMemcpyDeviceToDevice));
    }
    const double bytes = (double)n * 8;
    std::vector<Variant> vs;
    Geom g{nx, ny, 0, ny, 1, 0, 0};
    const float om = 0.5f;
    add_abl<64, true, true, true, 0>(vs, a, b, g, om, bytes, 8);
    add_abl<64, true, true, true, 2>(vs, a, b, g, om, bytes, 8);
    add_abl<64, true, true, true, 0>(vs, a, b, g, om, bytes, 32);
    add_abl<64, true, true, true, 2>(vs, a, b, g, om, bytes, 32);
    {   // ping-pong a<->b like the time loop does
        auto kern = k_pull_abl<float, 4, 64, true, true, true, 0>;
        const size_t lds = (160 * 1024 / 8) & ~255u;
        const dim3 grid(g.nx / 4 / 64, g.ny);
        vs.push_back({"abl b64 pingpong 8w/CU", 2 * bytes, [=] { hipLaunchKernelGGL(kern, grid, dim3(64), lds, 0, a, b, g, om); hipLaunchKernelGGL(kern, grid, dim3(64), lds, 0, b, a, g, om); }, {}});
    }
    product_variants(vs, a, b, g, om, bytes);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto &v : vs) v.launch();
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
    printf("%-34s %10s %10s %10s\n", "variant", "med ms", "med GB/s", "best GB/s");
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        printf("%-34s %10.3f %10.1f %10.1f\n", v.name.c_str(), v.ms[v.ms.size() / 2], v.bytes / v.ms[v.ms.size() / 2] / 1e6, v.bytes / v.ms[0] / 1e6);
    }
}
