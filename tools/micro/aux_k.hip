// Tuning harness (not product): launch shapes of the initialisation fill and of the max-|u| reduction
// at 16384 x 16384 fp32 / fp64 (both are one pass over nine planes: 36 / 72 bytes per node).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I lbm_amd/csrc -I include tools/micro/aux_k.hip -o tools/micro/aux_k
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#include "lbm_device.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2);} } while (0)
using namespace lbm;

template <typename T, int V, int BLOCK, bool NT>
__global__ void __launch_bounds__(BLOCK)
k_fill_vec(T *__restrict__ f, uint32_t nx, uint32_t ny, uint32_t y_first, uint32_t rows, const T *__restrict__ table)
{
    const uint32_t v = blockIdx.x * BLOCK + threadIdx.x;
    if (v >= nx / V) return;
    const size_t plane = (size_t)ny * nx;
    for (uint32_t r = blockIdx.y; r < rows; r += gridDim.y) {
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const T t = table[q * rows + r];
            typename VecTypes<T, V>::vec_t vec;
#pragma unroll
            for (int i = 0; i < V; ++i) vec[i] = t;
            typename VecTypes<T, V>::vec_a *p = reinterpret_cast<typename VecTypes<T, V>::vec_a *>(rowp(f, plane, q, y_first + r, nx) + (size_t)v * V);
            if (NT) __builtin_nontemporal_store(vec, p); else *p = vec;
        }
    }
}

template <typename F>
double time_ms(F launch, int reps = 5)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    launch();
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

template <typename T>
void run(uint32_t nx, uint32_t ny)
{
    constexpr int V = 16 / sizeof(T);
    const size_t n = (size_t)9 * nx * ny;
    T *f; CK(hipMalloc(&f, n * sizeof(T)));
    T *table; CK(hipMalloc(&table, 9 * (size_t)ny * sizeof(T)));
    std::vector<T> h(9 * (size_t)ny);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (T)(0.1 + 1e-6 * (double)(i % 1000));
    CK(hipMemcpy(table, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    uint32_t *key; CK(hipMalloc(&key, 256));
    const double gb = (double)n * sizeof(T) / 1e9;
    auto report = [&](const char *name, double ms) { printf("%-44s %7.3f ms  %6.0f GB/s\n", name, ms, gb / ms * 1e3); };
    char nm[128];
    const char *tn = sizeof(T) == 4 ? "f32" : "f64";
    // ---- fill
    {
        const unsigned gy = ny > 65535 ? 65535 : ny;
        snprintf(nm, sizeof nm, "%s fill scalar B256 (shipped)", tn);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_fill_rows<T, 256>), dim3((nx + 255) / 256, gy), dim3(256), 0, 0, f, nx, ny, 0u, ny, (const T *)table); }));
    }
    for (unsigned gy : {0u}) {
        const unsigned g = gy == 0 ? (ny > 65535 ? 65535 : ny) : gy;
        const uint32_t nvec = nx / V;
        snprintf(nm, sizeof nm, "%s fill vec B64 nt gy%u", tn, g);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_fill_vec<T, V, 64, true>), dim3((nvec + 63) / 64, g), dim3(64), 0, 0, f, nx, ny, 0u, ny, (const T *)table); }));
        snprintf(nm, sizeof nm, "%s fill vec B256 nt gy%u", tn, g);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_fill_vec<T, V, 256, true>), dim3((nvec + 255) / 256, g), dim3(256), 0, 0, f, nx, ny, 0u, ny, (const T *)table); }));
        snprintf(nm, sizeof nm, "%s fill vec B256 plain gy%u", tn, g);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_fill_vec<T, V, 256, false>), dim3((nvec + 255) / 256, g), dim3(256), 0, 0, f, nx, ny, 0u, ny, (const T *)table); }));
    }
    // ---- max |u|
    const uint32_t nvec = nx / V;
    for (unsigned gxcap : {64u})
        for (unsigned gycap : {1024u, 2048u, 4096u, 8192u, 0u}) {
            unsigned gx = (nvec + 255) / 256; if (gxcap && gx > gxcap) gx = gxcap;
            unsigned gy = gycap ? std::min<unsigned>(ny, gycap) : std::min<unsigned>(ny, 65535u);
            snprintf(nm, sizeof nm, "%s max B256 gx%u gy%u%s", tn, gx, gy, (gxcap == 64 && gycap == 1024) ? " (shipped)" : "");
            report(nm, time_ms([&] { CK(hipMemsetAsync(key, 0, 4, 0)); hipLaunchKernelGGL((k_max_speed<T, V, 256>), dim3(gx, gy), dim3(256), 0, 0, (const T *)f, nx, ny, 0u, ny, key); }));
            unsigned gx64 = (nvec + 63) / 64; if (gxcap && gx64 > gxcap * 4) gx64 = gxcap * 4;
            snprintf(nm, sizeof nm, "%s max B64  gx%u gy%u", tn, gx64, gy);
            report(nm, time_ms([&] { CK(hipMemsetAsync(key, 0, 4, 0)); hipLaunchKernelGGL((k_max_speed<T, V, 64>), dim3(gx64, gy), dim3(64), 0, 0, (const T *)f, nx, ny, 0u, ny, key); }));
        }
    CK(hipFree(f)); CK(hipFree(table)); CK(hipFree(key));
}

int main(int argc, char **argv)
{
    const uint32_t nx = argc > 1 ? atoi(argv[1]) : 16384, ny = argc > 2 ? atoi(argv[2]) : 16384;
    run<float>(nx, ny);
    run<double>(nx, ny);
    return 0;
}
