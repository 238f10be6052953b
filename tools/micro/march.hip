// Tuning harness (not product): the memory access pattern of the multi-step sweeps without their
// arithmetic.  A wavefront owns a strip of a row (64 lanes x LB bytes, W lanes stored, the window
// starting H lanes to the left) and marches it through a segment of rows; RB rows of nine planes are
// requested at once, then stored.  Reports the rate of 9 planes in + 9 planes out at nx x ny fp32.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/march.hip -o tools/micro/march
//   march nx ny  LB RB W H seg_rows group waves_per_cu [extra_rows [barrier [layout]]]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2);} } while (0)

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
extern __shared__ unsigned char lds_raw[];

struct Geom { uint32_t nx, ny, seg_rows, nstrips, nsegs, W, H, extra, barrier; size_t qs, ys; };

template <int LB> struct Vec;
template <> struct Vec<8> { typedef u32x2 t; };
template <> struct Vec<16> { typedef u32x4 t; };

template <int LB>
__device__ __forceinline__ typename Vec<LB>::t ld(const float *row, uint32_t bytes, uint32_t off)
{
    const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(row), 0, (int)bytes, 0x00020000);
    if constexpr (LB == 8) return __builtin_amdgcn_raw_buffer_load_b64(d, (int)off, 0, 0);
    else return __builtin_amdgcn_raw_buffer_load_b128(d, (int)off, 0, 0);
}
template <int LB>
__device__ __forceinline__ void st(float *row, uint32_t bytes, uint32_t off, typename Vec<LB>::t v)
{
    const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc(row, 0, (int)bytes, 0x00020000);
    if constexpr (LB == 8) __builtin_amdgcn_raw_buffer_store_b64(v, d, (int)off, 0, 2);
    else __builtin_amdgcn_raw_buffer_store_b128(v, d, (int)off, 0, 2);
}

template <int LB, int RB>
__global__ void __launch_bounds__(1024) k_march(const float *__restrict__ a, float *__restrict__ b, const Geom g)
{
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / 64));
    const uint32_t block = blockIdx.x * (blockDim.x / 64) + wave;
    const uint32_t strip = block % g.nstrips, seg = block / g.nstrips;
    if (seg >= g.nsegs) return;
    const uint32_t lane = threadIdx.x % 64;
    const uint32_t nxl = g.nx * 4 / LB;                       // lanes per row
    int64_t vx = (int64_t)strip * g.W - g.H + lane;
    vx %= (int64_t)nxl;
    if (vx < 0) vx += nxl;
    const uint32_t xoff = (uint32_t)vx * LB;
    const bool store_lane = lane >= g.H && lane < g.H + min(g.W, nxl - strip * g.W);
    const uint32_t soff = store_lane ? xoff : 0x80000000u;
    const uint32_t row_bytes = g.nx * 4;
    const uint32_t ys = seg * g.seg_rows, ye = min(ys + g.seg_rows, g.ny);
    // `extra` rows are read ahead of the segment and not stored (the y halo of the real kernels)
    for (int64_t r0 = (int64_t)ys - g.extra; r0 < (int64_t)ye; r0 += RB) {
        typename Vec<LB>::t v[RB][9];
        if (g.barrier) __builtin_amdgcn_s_barrier();   // keeps the wavefronts of a workgroup in step (equal trip counts only)
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            int64_t r = r0 + j;
            r = r < 0 ? r + g.ny : (r >= (int64_t)g.ny ? r - g.ny : r);
#pragma unroll
            for (int q = 0; q < 9; ++q) v[j][q] = ld<LB>(a + q * g.qs + (size_t)r * g.ys, row_bytes, xoff);
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int64_t r = r0 + j;
            const bool ok = r >= (int64_t)ys && r < (int64_t)ye;
            const uint32_t rr = ok ? (uint32_t)r : ys;
#pragma unroll
            for (int q = 0; q < 9; ++q) st<LB>(b + q * g.qs + (size_t)rr * g.ys, ok ? row_bytes : 0u, soff, v[j][q]);
        }
    }
}

template <int LB, int RB>
void run(uint32_t nx, uint32_t ny, const float *a, float *b, Geom g, int group, int wpc, const char *tag)
{
    const uint32_t nxl = nx * 4 / LB;
    g.nstrips = (nxl + g.W - 1) / g.W;
    g.nsegs = (ny + g.seg_rows - 1) / g.seg_rows;
    const uint32_t waves = g.nstrips * g.nsegs;
    const dim3 grid((waves + group - 1) / group);
    // residency cap: an unused LDS allocation per workgroup
    unsigned lds = wpc > 0 ? (160 * 1024 / (wpc / group > 0 ? wpc / group : 1)) & ~255u : 0;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void *)k_march<LB, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((k_march<LB, RB>), grid, dim3(64 * group), lds, 0, (i & 1) ? b : a, (i & 1) ? const_cast<float *>(a) : b, g);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t / 4);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    const double bytes = 2.0 * 9 * 4 * (double)nx * ny;
    printf("%-10s LB%-2d RB%d W%-2u H%u S%-4u G%d wpc%-2d x%u b%u L%d  waves %6u  med %6.3f ms  %6.0f GB/s (of the 18 planes)\n", tag, LB, RB, g.W, g.H,
           g.seg_rows, group, wpc, g.extra, g.barrier, g.qs == g.nx ? 1 : 0, waves, ms[ms.size() / 2], bytes / ms[ms.size() / 2] / 1e6);
}

int main(int argc, char **argv)
{
    if (argc < 10) { fprintf(stderr, "march nx ny LB RB W H seg_rows group waves_per_cu [extra_rows [barrier [layout]]]\n"); return 2; }
    const uint32_t nx = atoi(argv[1]), ny = atoi(argv[2]);
    const int LB = atoi(argv[3]), RB = atoi(argv[4]);
    Geom g{nx, ny, (uint32_t)atoi(argv[7]), 0, 0, (uint32_t)atoi(argv[5]), (uint32_t)atoi(argv[6]), argc > 10 ? (uint32_t)atoi(argv[10]) : 0u, argc > 11 ? (uint32_t)atoi(argv[11]) : 0u, 0, 0};
    const int layout = argc > 12 ? atoi(argv[12]) : 0;   // 0: [9][NY][NX]; 1: [NY][9][NX]
    g.qs = layout ? nx : (size_t)nx * ny;
    g.ys = layout ? (size_t)9 * nx : nx;
    const int group = atoi(argv[8]), wpc = atoi(argv[9]);
    const size_t n = (size_t)9 * nx * ny;
    float *a, *b;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    CK(hipMemset(a, 0x11, n * 4)); CK(hipMemset(b, 0x22, n * 4));
    const char *tag = getenv("TAG") ? getenv("TAG") : "march";
#define CASE(lb, rb) if (LB == lb && RB == rb) { run<lb, rb>(nx, ny, a, b, g, group, wpc, tag); return 0; }
    CASE(8, 1) CASE(8, 2) CASE(8, 4) CASE(8, 8) CASE(16, 1) CASE(16, 2) CASE(16, 4)
    fprintf(stderr, "unsupported LB/RB\n");
    return 2;
}
