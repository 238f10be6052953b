// Tuning harness (not product): times the product fused kernel for the lid-driven cavity (fp64, or
// -DTUNE_T=float -DTUNE_V=2) with the ring / prefetch macros given on the compiler command line.  No result check: the product tests
// cover whatever setting is adopted.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "lbm_device.h"
#include "lbm_fused.h"
#pragma clang fp contract(off)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2);} } while (0)
using namespace lbm;

template <typename T, int V, int K, bool PUSH, bool LID>
void run(T *a, T *b, uint32_t nx, uint32_t ny, uint32_t seg_rows)
{
    constexpr int W = 64 - 2 * fused_halo_lanes<T, V, K>();
    FusedGeom g{nx, ny, 0, ny, LID ? 0 : 1, seg_rows, (nx / V + W - 1) / W, (ny + seg_rows - 1) / seg_rows, 0, 0, ny};
    auto kern = k_fused<T, V, K, PUSH, LID>;
    const dim3 grid(g.nstrips * g.nsegs);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, dim3(64), 0, 0, a, b, g, (T)0.5, (T)(0.1 / 6));
    CK(hipEventRecord(e0));
    const int n = 5;
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kern, grid, dim3(64), 0, 0, a, b, g, (T)0.5, (T)(0.1 / 6));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= n;
    printf("f%d K%d %s %s S%u: %.3f ms %.1f GLUPS\n", (int)sizeof(T) * 8, K, PUSH ? "push" : "pull", LID ? "lid" : "periodic",
           seg_rows, ms, (double)nx * ny * K / ms / 1e6);
    fflush(stdout);
}

template <typename T> __global__ void k_fill(T *p, size_t n, T v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

#ifndef TUNE_T
#define TUNE_T double
#define TUNE_V 1
#endif

int main(int argc, char **argv)
{
    const uint32_t nx = argc > 1 ? atoi(argv[1]) : 16384, ny = argc > 2 ? atoi(argv[2]) : 16384;
    const uint32_t seg = argc > 3 ? atoi(argv[3]) : 128;
    typedef TUNE_T T;
    constexpr int V = TUNE_V;
    const size_t n = (size_t)9 * nx * ny;
    T *a, *b;
    CK(hipMalloc(&a, n * sizeof(T))); CK(hipMalloc(&b, n * sizeof(T)));
    hipLaunchKernelGGL(k_fill<T>, dim3(4096), dim3(256), 0, 0, a, n, (T)(1.0 / 9));
    CK(hipDeviceSynchronize());
    run<T, V, 4, false, true>(a, b, nx, ny, seg);
    run<T, V, 4, true, true>(a, b, nx, ny, seg);
    run<T, V, 5, false, true>(a, b, nx, ny, seg);
    run<T, V, 5, true, true>(a, b, nx, ny, seg);
    run<T, V, 6, false, true>(a, b, nx, ny, seg);
    run<T, V, 6, true, true>(a, b, nx, ny, seg);
    run<T, V, 6, false, false>(a, b, nx, ny, seg);
    run<T, V, 6, true, false>(a, b, nx, ny, seg);
    return 0;
}
