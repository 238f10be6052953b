// Tuning harness (not product): times lbm::k_fused<float, 2, K, PUSH, false, false, ARITH> at 16384^2
// for the macro settings this file is compiled with, and checks the result against K launches of
// the one-step instantiation of the same arithmetic.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I lbm_amd/csrc -I include \
//         -DTK_ARITH=2 -DTK_KS="6,7,8" [-DLBM_FUSED_PREFETCH_F32=2 ...] tools/micro/tune_k.hip -o tune_k_x
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
#ifndef TK_ARITH
#define TK_ARITH 2
#endif
#ifndef TK_PUSH
#define TK_PUSH false
#endif
#ifndef TK_T
#define TK_T float
#endif
typedef TK_T T;
constexpr int V = kFusedLaneBytes / (int)sizeof(T);

__global__ void k_diff(const T *a, const T *b, size_t n, unsigned long long *cnt)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (; i < n; i += stride) {
        typedef typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type U;
        if (*reinterpret_cast<const U *>(&a[i]) != *reinterpret_cast<const U *>(&b[i])) ++c;
    }
    if (c) atomicAdd(cnt, c);
}

struct Variant { std::string name; int K; std::function<void(const T *, T *)> launch; std::vector<float> ms; };

template <int K>
void add(std::vector<Variant> &vs, uint32_t nx, uint32_t ny, T om, uint32_t seg_rows)
{
    constexpr int W = 64 - 2 * fused_halo_lanes<T, V, K>();
    FusedGeom g{nx, ny, 0, ny, 1, seg_rows, (nx / V + W - 1) / W, (ny + seg_rows - 1) / seg_rows, 0, 0, 0, 0, 0, 0};
    constexpr unsigned NW = (unsigned)fused_group_waves<T, V, K, TK_PUSH, false, false, TK_ARITH>();
    constexpr unsigned lds = NW * (unsigned)fused_lds_bytes_per_wave<T, V, K, TK_PUSH, false, false, TK_ARITH>();
    const dim3 grid((g.nstrips * g.nsegs + NW - 1) / NW);
    auto kern = k_fused<T, V, K, TK_PUSH, false, false, TK_ARITH>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    char nm[96];
    snprintf(nm, sizeof nm, "K%d_S%u", K, seg_rows);
    vs.push_back({nm, K, [=](const T *a, T *b) { hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, 0, a, b, g, om, (T)0); }, {}});
}

// segment length in 96..128 that minimises rounds x iterations (as the library does)
uint32_t pick_seg(uint32_t nx, uint32_t ny, int K, uint64_t slots)
{
    const int W = 64 - 2 * (K <= 4 ? fused_halo_lanes<T, V, 4>() : fused_halo_lanes<T, V, 6>());
    const uint64_t nstrips = (nx / V + W - 1) / W;
    const uint64_t extra = TK_PUSH ? 2 * K : 2 * (K - 1);
    uint32_t best = 128; double bc = 1e30;
    for (uint32_t s = 96; s <= 128; ++s) {
        const uint64_t blocks = nstrips * ((ny + s - 1) / s);
        const double cost = (double)((blocks + slots - 1) / slots) * (s + extra);
        if (cost < bc) { bc = cost; best = s; }
    }
    return best;
}

int main(int argc, char **argv)
{
    uint32_t nx = 16384, ny = 16384; int rounds = 3, iters = 6;
    if (argc > 1) nx = atoi(argv[1]);
    if (argc > 2) ny = atoi(argv[2]);
    if (getenv("TK_ITERS")) iters = atoi(getenv("TK_ITERS"));     // launches per timed burst
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
    const int ks[] = {TK_KS};
    std::vector<uint32_t> segs;   // segment lengths to try (default: the library's choice)
    for (int a = 3; a < argc; ++a) segs.push_back((uint32_t)atoi(argv[a]));
    for (int K : ks) {
        for (uint32_t S : segs.empty() ? std::vector<uint32_t>{pick_seg(nx, ny, K, 2048)} : segs)
        switch (K) {
            case 1: add<1>(vs, nx, ny, om, S); break;
            case 2: add<2>(vs, nx, ny, om, S); break;
            case 3: add<3>(vs, nx, ny, om, S); break;
            case 4: add<4>(vs, nx, ny, om, S); break;
            case 5: add<5>(vs, nx, ny, om, S); break;
            case 6: add<6>(vs, nx, ny, om, S); break;
            case 7: add<7>(vs, nx, ny, om, S); break;
            case 8: add<8>(vs, nx, ny, om, S); break;
        }
    }
    std::vector<Variant> one;
    add<1>(one, nx, ny, om, 128);
    unsigned long long bad = 0;
    for (auto &v : vs) {
        const T *src = a; T *dst = c, *other = d;
        for (int k = 0; k < v.K; ++k) { one[0].launch(src, dst); src = dst; std::swap(dst, other); }
        CK(hipMemset(b, 0xff, n * sizeof(T)));
        v.launch(a, b);
        CK(hipMemset(cnt, 0, 8));
        hipLaunchKernelGGL(k_diff, dim3(4096), dim3(256), 0, 0, src, b, n, cnt);
        unsigned long long hc; CK(hipMemcpy(&hc, cnt, 8, hipMemcpyDeviceToHost));
        printf("check %-12s mismatches %llu\n", v.name.c_str(), hc);
        bad += hc;
    }
    // developed data for the timing: a few steps from the synthetic start
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; ++r)
        for (auto &v : vs) {
            CK(hipEventRecord(e0, 0));
            const T *src = a; T *dst = b;
            for (int i = 0; i < iters; ++i) { v.launch(src, dst); T *t = const_cast<T *>(src); src = dst; dst = (t == a && i == 0) ? c : t; }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms / iters);
        }
    CK(hipGetLastError());
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        const double lups = (double)nx * ny * v.K;
        printf("%-14s %-12s med %7.3f ms %8.0f MLUPS   best %8.0f\n", TK_TAG, v.name.c_str(), v.ms[v.ms.size() / 2], lups / v.ms[v.ms.size() / 2] / 1e3, lups / v.ms[0] / 1e3);
    }
    return bad ? 1 : 0;
}
