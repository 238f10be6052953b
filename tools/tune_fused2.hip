// Tuning harness (not product): the product fused kernel (lbm_fused.h) with other vector widths /
// halo widths / segment lengths, checked bit for bit against K launches of the one-step kernel.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include "lbm_device.h"
#include "lbm_fused.h"
#pragma clang fp contract(off)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2);} } while (0)
using namespace lbm;

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

struct Variant { std::string name; int K; double lups; std::function<void()> launch; std::vector<float> ms; };

template <typename T, int V, int K, bool PUSH>
void add_fused(std::vector<Variant> &vs, T *a, T *b, uint32_t nx, uint32_t ny, T om, uint32_t seg_rows)
{
    constexpr int W = 64 - 2 * fused_halo_lanes<T, V, K>();
    FusedGeom g{nx, ny, 0, ny, 1, seg_rows, (nx / V + W - 1) / W, (ny + seg_rows - 1) / seg_rows, 0, 0, 0};
    auto kern = k_fused<T, V, K, PUSH, false>;
    const dim3 grid(g.nstrips * g.nsegs);
    char nm[96];
    snprintf(nm, sizeof nm, "K%d_%s_f%d_V%d_S%u", K, PUSH ? "push" : "pull", (int)sizeof(T) * 8, V, seg_rows);
    vs.push_back({nm, K, (double)nx * ny * K, [=] { hipLaunchKernelGGL(kern, grid, dim3(64), 0, 0, a, b, g, om, (T)0); }, {}});
}

template <typename T, int V, bool PUSH>
void product(T *a, T *b, uint32_t nx, uint32_t ny, T om)
{
    Geom g{0, nx, ny, 0, ny, 1, 0, 0};
    const dim3 grid(nx / V / 64 + ((nx / V) % 64 ? 1 : 0), ny, 1);
    const size_t lds = (160 * 1024 / 6) & ~255u;
    if (PUSH) {
        auto kern = k_push_periodic_vec<T, V, 64>;
        CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(kern, grid, dim3(64), lds, 0, a, b, g, om);
    } else {
        auto kern = k_pull_periodic_vec<T, V, 64>;
        CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(kern, grid, dim3(64), lds, 0, a, b, g, om);
    }
}

template <typename T, int VP, bool PUSH>
int run(uint32_t nx, uint32_t ny, int rounds, int iters, const char *only)
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
    constexpr int V2 = VP / 2;
    for (uint32_t S : {128u, 192u}) {
        add_fused<T, V2, 4, PUSH>(vs, a, b, nx, ny, om, S);
        add_fused<T, V2, 5, PUSH>(vs, a, b, nx, ny, om, S);
        add_fused<T, V2, 6, PUSH>(vs, a, b, nx, ny, om, S);
        if constexpr (sizeof(T) == 8) {
            add_fused<T, V2, 7, PUSH>(vs, a, b, nx, ny, om, S);
            add_fused<T, V2, 8, PUSH>(vs, a, b, nx, ny, om, S);
        }
    }
    if (only) {
        std::vector<Variant> keep;
        for (auto &v : vs) if (strstr(v.name.c_str(), only)) keep.push_back(v);
        vs.swap(keep);
    }
    unsigned long long bad = 0;
    if (!only)
        for (auto &v : vs) {
            T *src = a, *dst = c, *other = d;
            for (int k = 0; k < v.K; ++k) { product<T, VP, PUSH>(src, dst, nx, ny, om); src = dst; std::swap(dst, other); }
            CK(hipMemset(b, 0xff, n * sizeof(T)));
            v.launch();
            CK(hipMemset(cnt, 0, 8));
            hipLaunchKernelGGL(k_diff<T>, dim3(4096), dim3(256), 0, 0, src, b, n, cnt);
            unsigned long long hc; CK(hipMemcpy(&hc, cnt, 8, hipMemcpyDeviceToHost));
            printf("check %-36s mismatches %llu of %zu\n", v.name.c_str(), hc, n);
            bad += hc;
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
    uint32_t nx = 16384, ny = 16384; int rounds = 3, iters = 8;
    const char *prec = "f32", *scheme = "pull";
    if (argc > 1) nx = atoi(argv[1]);
    if (argc > 2) ny = atoi(argv[2]);
    if (argc > 3) prec = argv[3];
    if (argc > 4) scheme = argv[4];
    if (argc > 5) iters = atoi(argv[5]);
    const char *only = argc > 6 ? argv[6] : nullptr;
    if (only) rounds = 1;
    const bool push = !strcmp(scheme, "push");
    if (!strcmp(prec, "f64")) return push ? run<double, 2, true>(nx, ny, rounds, iters, only) : run<double, 2, false>(nx, ny, rounds, iters, only);
    return push ? run<float, 4, true>(nx, ny, rounds, iters, only) : run<float, 4, false>(nx, ny, rounds, iters, only);
}
