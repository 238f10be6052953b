// Copy-ceiling exploration (tuning only): which launch shape gets the most HBM read+write
// bandwidth for 1:1 streams on MI355X.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2);} } while (0)

typedef float v4 __attribute__((ext_vector_type(4)));

// one-shot: block tile = BLOCK*U vectors, thread handles U vectors strided by BLOCK
template <int BLOCK, int U, int NT>
__global__ void __launch_bounds__(BLOCK) k_copy_tile(const v4 *__restrict__ a, v4 *__restrict__ b, size_t n)
{
    const size_t base = (size_t)blockIdx.x * (BLOCK * U) + threadIdx.x;
    v4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * BLOCK;
        if (i < n) r[u] = (NT & 2) ? __builtin_nontemporal_load(a + i) : a[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const size_t i = base + (size_t)u * BLOCK;
        if (i < n) { if (NT & 1) __builtin_nontemporal_store(r[u], b + i); else b[i] = r[u]; }
    }
}

// 9 planes: thread handles the same vector index in P planes (stride plane), rows via blockIdx.y
template <int BLOCK, int P, int NT>
__global__ void __launch_bounds__(BLOCK) k_copy_planes(const v4 *__restrict__ a, v4 *__restrict__ b, size_t plane_v, uint32_t row_v, uint32_t rows)
{
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

// read-only and write-only probes
template <int BLOCK, int U>
__global__ void __launch_bounds__(BLOCK) k_read(const v4 *__restrict__ a, v4 *__restrict__ b, size_t n)
{
    const size_t base = (size_t)blockIdx.x * (BLOCK * U) + threadIdx.x;
    v4 s = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * BLOCK; if (i < n) s += a[i]; }
    if (s.x == 123.456f) b[base] = s;
}
template <int BLOCK, int U>
__global__ void __launch_bounds__(BLOCK) k_write(v4 *__restrict__ b, size_t n)
{
    const size_t base = (size_t)blockIdx.x * (BLOCK * U) + threadIdx.x;
    v4 s = {1, 2, 3, 4};
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * BLOCK; if (i < n) b[i] = s; }
}

struct Variant { std::string name; double bytes; std::function<void()> launch; std::vector<float> ms; };

int main(int argc, char **argv)
{
    uint32_t nx = 16384, ny = 16384; int rounds = 5, iters = 5;
    if (argc > 1) nx = atoi(argv[1]);
    if (argc > 2) ny = atoi(argv[2]);
    const size_t n = (size_t)9 * nx * ny, nv = n / 4;
    v4 *a, *b;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    CK(hipMemset(a, 0x3c, n * 4)); CK(hipMemset(b, 0, n * 4));
    const double bytes = (double)n * 8;
    std::vector<Variant> vs;
#define TILE(B, U, NT) vs.push_back({"tile b" #B " u" #U " nt" #NT, bytes, [&] { hipLaunchKernelGGL((k_copy_tile<B, U, NT>), dim3((unsigned)((nv + (size_t)B * U - 1) / ((size_t)B * U))), dim3(B), 0, 0, a, b, nv); }, {}})
    TILE(256, 1, 0); TILE(256, 2, 0); TILE(256, 4, 0); TILE(256, 8, 0); TILE(256, 9, 0);
    TILE(64, 4, 0); TILE(64, 9, 0); TILE(128, 4, 0); TILE(512, 4, 0); TILE(1024, 4, 0);
    TILE(256, 4, 1); TILE(256, 4, 2); TILE(256, 4, 3); TILE(64, 9, 3); TILE(256, 9, 3);
    const size_t plane_v = (size_t)nx * ny / 4; const uint32_t row_v = nx / 4;
#define PLANES(B, NT, GY) vs.push_back({"planes9 b" #B " nt" #NT " gy" #GY, bytes, [&] { hipLaunchKernelGGL((k_copy_planes<B, 9, NT>), dim3((row_v + B - 1) / B, GY), dim3(B), 0, 0, a, b, plane_v, row_v, ny); }, {}})
    PLANES(256, 0, 16384); PLANES(64, 0, 16384); PLANES(128, 0, 16384); PLANES(64, 3, 16384); PLANES(128, 3, 16384); PLANES(256, 3, 16384); PLANES(512, 3, 16384); PLANES(1024, 3, 16384);
    PLANES(256, 3, 8192); PLANES(256, 3, 4096);
    vs.push_back({"read  b256 u4", bytes / 2, [&] { hipLaunchKernelGGL((k_read<256, 4>), dim3((unsigned)((nv + 1023) / 1024)), dim3(256), 0, 0, a, b, nv); }, {}});
    vs.push_back({"read  b256 u9", bytes / 2, [&] { hipLaunchKernelGGL((k_read<256, 9>), dim3((unsigned)((nv + 2303) / 2304)), dim3(256), 0, 0, a, b, nv); }, {}});
    vs.push_back({"write b256 u4", bytes / 2, [&] { hipLaunchKernelGGL((k_write<256, 4>), dim3((unsigned)((nv + 1023) / 1024)), dim3(256), 0, 0, b, nv); }, {}});

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto &v : vs) v.launch();
    CK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; ++r)
        for (auto &v : vs) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) v.launch();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms / iters);
        }
    CK(hipGetLastError());
    printf("%-32s %10s %10s %10s\n", "variant", "med ms", "med GB/s", "best GB/s");
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        printf("%-32s %10.3f %10.1f %10.1f\n", v.name.c_str(), v.ms[v.ms.size() / 2], v.bytes / v.ms[v.ms.size() / 2] / 1e6, v.bytes / v.ms[0] / 1e6);
    }
}
