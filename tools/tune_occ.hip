// Occupancy sweep (tuning only): the 9-plane copy and the pull sweep at limited waves per CU,
// limited by dynamic LDS size.
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
typedef float v4 __attribute__((ext_vector_type(4)));

template <int BLOCK, int P, int NT>
__global__ void __launch_bounds__(BLOCK) k_copy_planes(const v4 *__restrict__ a, v4 *__restrict__ b, size_t plane_v, uint32_t row_v, uint32_t rows)
{
    extern __shared__ char lds[];
    if (row_v == 0xffffffffu) lds[threadIdx.x] = 1;  // keep the allocation
    const uint32_t xv = blockIdx.x * BLOCK + threadIdx.x;
    if (xv >= row_v) return;
    for (uint32_t y = blockIdx.y; y < rows; y += gridDim.y) {
        v4 r[P];
        const size_t o = (size_t)y * row_v + xv;
#pragma unroll
        for (int q = 0; q < P; ++q) r[q] = (NT & 2) ? __builtin_nontemporal_load(a + q * plane_v + o) : a[q * plane_v + o];
#pragma unroll
        for (int q = 0; q < P; ++q) { if (NT & 1) __builtin_nontemporal_store(r[q], b + q * plane_v + o); else b[q * plane_v + o] = r[q]; }
    }
}

template <typename T, int V, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_pull_lds(const T *__restrict__ f, T *__restrict__ buf, const Geom g, const T om)
{
    extern __shared__ char lds[];
    if (g.nx == 0xffffffffu) lds[threadIdx.x] = 1;
    const uint32_t x0 = (blockIdx.x * BLOCK + threadIdx.x) * V;
    if (x0 >= g.nx) return;
    const size_t plane = (size_t)g.ny * g.nx;
    for (uint32_t y = g.y_lo + blockIdx.y; y < g.y_hi; y += gridDim.y) {
        const uint32_t yu = row_up(g, y), yd = row_down(g, y);
        T c0[V], c1[V], c2[V], c3[V], c4[V], c5[V], c6[V], c7[V], c8[V];
        load_from_right<T, V>(c7, rowp(f, plane, 7, yu, g.nx), x0, g.nx);
        load_a<T, V>(c4, rowp(f, plane, 4, yu, g.nx) + x0);
        load_from_left<T, V>(c8, rowp(f, plane, 8, yu, g.nx), x0, g.nx);
        load_from_right<T, V>(c3, rowp(f, plane, 3, y, g.nx), x0, g.nx);
        load_a<T, V>(c0, rowp(f, plane, 0, y, g.nx) + x0);
        load_from_left<T, V>(c1, rowp(f, plane, 1, y, g.nx), x0, g.nx);
        load_from_right<T, V>(c6, rowp(f, plane, 6, yd, g.nx), x0, g.nx);
        load_a<T, V>(c2, rowp(f, plane, 2, yd, g.nx) + x0);
        load_from_left<T, V>(c5, rowp(f, plane, 5, yd, g.nx), x0, g.nx);
        T o[9][V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
            T r[9];
            collide<T>(c0[i], c1[i], c2[i], c3[i], c4[i], c5[i], c6[i], c7[i], c8[i], om, r);
#pragma unroll
            for (int q = 0; q < 9; ++q) o[q][i] = r[q];
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) store_a<T, V>(rowp(buf, plane, q, y, g.nx) + x0, o[q]);
    }
}

struct Variant { std::string name; double bytes; std::function<void()> launch; std::vector<float> ms; };

int main(int argc, char **argv)
{
    uint32_t nx = 16384, ny = 16384; int rounds = 5, iters = 5;
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
    const size_t plane_v = (size_t)nx * ny / 4; const uint32_t row_v = nx / 4;
    Geom g{nx, ny, 0, ny, 1, 0, 0};
    const float om = 0.5f;
    CK(hipFuncSetAttribute((const void *)k_copy_planes<64, 9, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_copy_planes<64, 9, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_copy_planes<256, 9, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_pull_lds<float, 4, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_pull_lds<float, 4, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // waves per CU for 64-thread blocks = floor(160 KiB / lds)
    const int wpcs[] = {4, 8, 12, 16, 20, 24, 32};
    for (int wpc : wpcs) {
        const size_t lds = (wpc >= 32) ? 0 : (160 * 1024 / wpc) & ~255u;
        char nm[96];
        snprintf(nm, sizeof nm, "planes9 b64 nt0 %2d waves/CU", wpc);
        vs.push_back({nm, bytes, [=] { hipLaunchKernelGGL((k_copy_planes<64, 9, 0>), dim3((row_v + 63) / 64, ny), dim3(64), lds, 0, (const v4 *)a, (v4 *)b, plane_v, row_v, ny); }, {}});
        snprintf(nm, sizeof nm, "planes9 b64 nt3 %2d waves/CU", wpc);
        vs.push_back({nm, bytes, [=] { hipLaunchKernelGGL((k_copy_planes<64, 9, 3>), dim3((row_v + 63) / 64, ny), dim3(64), lds, 0, (const v4 *)a, (v4 *)b, plane_v, row_v, ny); }, {}});
        snprintf(nm, sizeof nm, "pull    b64     %2d waves/CU", wpc);
        vs.push_back({nm, bytes, [=] { hipLaunchKernelGGL((k_pull_lds<float, 4, 64>), dim3((row_v + 63) / 64, ny), dim3(64), lds, 0, a, b, g, om); }, {}});
    }
    // 256-thread blocks: blocks per CU = floor(160 KiB / lds), waves = 4x
    for (int bpc : {1, 2, 3, 4, 5}) {
        const size_t lds = (160 * 1024 / bpc) & ~255u;
        char nm[96];
        snprintf(nm, sizeof nm, "pull    b256    %2d waves/CU", bpc * 4);
        vs.push_back({nm, bytes, [=] { hipLaunchKernelGGL((k_pull_lds<float, 4, 256>), dim3((row_v + 255) / 256, ny), dim3(256), lds, 0, a, b, g, om); }, {}});
        snprintf(nm, sizeof nm, "planes9 b256 nt3 %2d waves/CU", bpc * 4);
        vs.push_back({nm, bytes, [=] { hipLaunchKernelGGL((k_copy_planes<256, 9, 3>), dim3((row_v + 255) / 256, ny), dim3(256), lds, 0, (const v4 *)a, (v4 *)b, plane_v, row_v, ny); }, {}});
    }

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
