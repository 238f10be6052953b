// Straight-line pull (one 64-lane chunk of one row per wave): cache-policy bits on loads/stores via
// buffer instructions, and waves/CU.  Tuning only.
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
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int AUX>
__device__ __forceinline__ void bload(float (&r)[4], __amdgpu_buffer_rsrc_t rs, uint32_t byte_off)
{
    v4i v = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, AUX));
    v4f w = __builtin_bit_cast(v4f, v);
    r[0] = w.x; r[1] = w.y; r[2] = w.z; r[3] = w.w;
}
template <int AUX>
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off, const float (&r)[4])
{
    v4f w = {r[0], r[1], r[2], r[3]};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, w), rs, byte_off, 0, AUX);
}

template <int LAUX, int SAUX>
__global__ void __launch_bounds__(64)
k_pull_buf(const float *__restrict__ f, float *__restrict__ buf, const Geom g, const float om)
{
    extern __shared__ char lds[];
    if (g.nx == 0xffffffffu) lds[threadIdx.x] = 1;
    const uint32_t x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const size_t plane = (size_t)g.ny * g.nx;
    const uint32_t y = g.y_lo + blockIdx.y;
    const uint32_t yu = row_up(g, y), yd = row_down(g, y);
    const uint32_t row_bytes = g.nx * 4;
    float c[9][4];
    // one descriptor per (plane,row): out-of-range lanes read 0 and are patched below
    auto rs = [&](const float *base, int q, uint32_t yy) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(base + ((size_t)q * plane + (size_t)yy * g.nx)), 0, row_bytes, 0x00020000);
    };
    const bool first = x0 == 0, last = x0 + 4 == g.nx;
    if (!first && !last) {
        bload<LAUX>(c[7], rs(f, 7, yu), (x0 + 1) * 4);
        bload<LAUX>(c[4], rs(f, 4, yu), x0 * 4);
        bload<LAUX>(c[8], rs(f, 8, yu), (x0 - 1) * 4);
        bload<LAUX>(c[3], rs(f, 3, y), (x0 + 1) * 4);
        bload<LAUX>(c[0], rs(f, 0, y), x0 * 4);
        bload<LAUX>(c[1], rs(f, 1, y), (x0 - 1) * 4);
        bload<LAUX>(c[6], rs(f, 6, yd), (x0 + 1) * 4);
        bload<LAUX>(c[2], rs(f, 2, yd), x0 * 4);
        bload<LAUX>(c[5], rs(f, 5, yd), (x0 - 1) * 4);
    } else {
        load_from_right<float, 4>(c[7], rowp(f, plane, 7, yu, g.nx), x0, g.nx);
        load_a<float, 4>(c[4], rowp(f, plane, 4, yu, g.nx) + x0);
        load_from_left<float, 4>(c[8], rowp(f, plane, 8, yu, g.nx), x0, g.nx);
        load_from_right<float, 4>(c[3], rowp(f, plane, 3, y, g.nx), x0, g.nx);
        load_a<float, 4>(c[0], rowp(f, plane, 0, y, g.nx) + x0);
        load_from_left<float, 4>(c[1], rowp(f, plane, 1, y, g.nx), x0, g.nx);
        load_from_right<float, 4>(c[6], rowp(f, plane, 6, yd, g.nx), x0, g.nx);
        load_a<float, 4>(c[2], rowp(f, plane, 2, yd, g.nx) + x0);
        load_from_left<float, 4>(c[5], rowp(f, plane, 5, yd, g.nx), x0, g.nx);
    }
    float o[9][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float r[9];
        collide<float>(c[0][i], c[1][i], c[2][i], c[3][i], c[4][i], c[5][i], c[6][i], c[7][i], c[8][i], om, r);
#pragma unroll
        for (int q = 0; q < 9; ++q) o[q][i] = r[q];
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) bstore<SAUX>(rs(buf, q, y), x0 * 4, o[q]);
}

struct Variant { std::string name; double bytes; std::function<void()> launch; std::vector<float> ms; };

template <int LAUX, int SAUX>
void add(std::vector<Variant> &vs, float *a, float *b, Geom g, float om, double bytes, int wpc)
{
    auto kern = k_pull_buf<LAUX, SAUX>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds = (wpc >= 32) ? 0 : (160 * 1024 / wpc) & ~255u;
    char nm[96];
    snprintf(nm, sizeof nm, "buf laux%-2d saux%-2d %2dw/CU", LAUX, SAUX, wpc);
    const dim3 grid(g.nx / 4 / 64, g.ny);
    vs.push_back({nm, bytes, [=] { hipLaunchKernelGGL(kern, grid, dim3(64), lds, 0, a, b, g, om); }, {}});
}

int main(int argc, char **argv)
{
    uint32_t nx = 16384, ny = 16384; int rounds = 5, iters = 4;
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
    for (int wpc : {6, 7, 8, 9, 10, 12}) add<0, 2>(vs, a, b, g, om, bytes, wpc);
    for (int wpc : {8}) {
        add<0, 0>(vs, a, b, g, om, bytes, wpc);
        add<0, 1>(vs, a, b, g, om, bytes, wpc);
        add<0, 3>(vs, a, b, g, om, bytes, wpc);
        add<0, 16>(vs, a, b, g, om, bytes, wpc);
        add<0, 17>(vs, a, b, g, om, bytes, wpc);
        add<0, 18>(vs, a, b, g, om, bytes, wpc);
        add<0, 19>(vs, a, b, g, om, bytes, wpc);
        add<1, 2>(vs, a, b, g, om, bytes, wpc);
        add<2, 2>(vs, a, b, g, om, bytes, wpc);
        add<16, 2>(vs, a, b, g, om, bytes, wpc);
        add<17, 2>(vs, a, b, g, om, bytes, wpc);
        add<18, 2>(vs, a, b, g, om, bytes, wpc);
        add<16, 18>(vs, a, b, g, om, bytes, wpc);
        add<17, 19>(vs, a, b, g, om, bytes, wpc);
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
    printf("%-30s %10s %10s %10s\n", "variant", "med ms", "med GB/s", "best GB/s");
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        printf("%-30s %10.3f %10.1f %10.1f\n", v.name.c_str(), v.ms[v.ms.size() / 2], v.bytes / v.ms[v.ms.size() / 2] / 1e6, v.bytes / v.ms[0] / 1e6);
    }
}
