// Issue-cost micro-benchmark for gfx950: what do packed-fp32 chains, the wait state the compiler
// puts between dependent packed operations, scalar instructions and a second/third wavefront per
// SIMD cost?  Each kernel repeats a 64-instruction body; cycles per instruction slot are reported
// relative to the wall clock of a dependent v_fma_f32 chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float float2_t __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

// A: dependent packed chain with the wait state
__global__ void k_pk_dep_nop(float *out, int n, float s)
{
    float2_t a = {s, s + 1.f}, b = {1.0001f, 0.9999f}, c = {1e-7f, 2e-7f};
    for (int i = 0; i < n; ++i) {
        REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0" : "+v"(a) : "v"(b), "v"(c));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.x + a.y;
}
// B: four independent packed chains interleaved, no wait states
__global__ void k_pk_indep(float *out, int n, float s)
{
    float2_t a0 = {s, s + 1.f}, a1 = {s + 2.f, s}, a2 = {s, s - 1.f}, a3 = {s + 3.f, s}, b = {1.0001f, 0.9999f}, c = {1e-7f, 2e-7f};
    for (int i = 0; i < n; ++i) {
        REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a0.y + a1.x + a1.y + a2.x + a2.y + a3.x + a3.y;
}
// B2: two independent packed chains interleaved (distance 2), no wait states
__global__ void k_pk_indep2(float *out, int n, float s)
{
    float2_t a0 = {s, s + 1.f}, a1 = {s + 2.f, s}, b = {1.0001f, 0.9999f}, c = {1e-7f, 2e-7f};
    for (int i = 0; i < n; ++i) {
        REP16(asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3"
                           : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a0.y + a1.x + a1.y;
}
// C: dependent scalar-fp32 chain (no wait state needed)
__global__ void k_f32_dep(float *out, int n, float s)
{
    float a = s, b = 1.0001f, c = 1e-7f;
    for (int i = 0; i < n; ++i) {
        REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
}
// D: four independent packed chains, one scalar add after every packed instruction
__global__ void k_pk_indep_salu(float *out, int n, float s)
{
    float2_t a0 = {s, s + 1.f}, a1 = {s + 2.f, s}, a2 = {s, s - 1.f}, a3 = {s + 3.f, s}, b = {1.0001f, 0.9999f}, c = {1e-7f, 2e-7f};
    int t = n;
    for (int i = 0; i < n; ++i) {
        REP16(asm volatile("v_pk_fma_f32 %0, %0, %5, %6\n s_add_i32 %4, %4, 1\n v_pk_fma_f32 %1, %1, %5, %6\n s_add_i32 %4, %4, 1\n v_pk_fma_f32 %2, %2, %5, %6\n s_add_i32 %4, %4, 1\n v_pk_fma_f32 %3, %3, %5, %6\n s_add_i32 %4, %4, 1"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(t) : "v"(b), "v"(c) : "scc");)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a0.y + a1.x + a1.y + a2.x + a2.y + a3.x + a3.y + (float)t;
}
// E: four independent packed chains, one s_nop 0 after every packed instruction
__global__ void k_pk_indep_nop(float *out, int n, float s)
{
    float2_t a0 = {s, s + 1.f}, a1 = {s + 2.f, s}, a2 = {s, s - 1.f}, a3 = {s + 3.f, s}, b = {1.0001f, 0.9999f}, c = {1e-7f, 2e-7f};
    for (int i = 0; i < n; ++i) {
        REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n s_nop 0\n v_pk_fma_f32 %1, %1, %4, %5\n s_nop 0\n v_pk_fma_f32 %2, %2, %4, %5\n s_nop 0\n v_pk_fma_f32 %3, %3, %4, %5\n s_nop 0"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a0.y + a1.x + a1.y + a2.x + a2.y + a3.x + a3.y;
}
// F: independent scalar-fp32 chains (4), for the plain VALU rate
__global__ void k_f32_indep(float *out, int n, float s)
{
    float a0 = s, a1 = s + 1.f, a2 = s + 2.f, a3 = s + 3.f, b = 1.0001f, c = 1e-7f;
    for (int i = 0; i < n; ++i) {
        REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
// G: v_mov_b32 with a DPP wavefront shift, independent (what the streaming step issues)
__global__ void k_dpp_mov(float *out, int n, float s)
{
    float a0 = s, a1 = s + 1.f, a2 = s + 2.f, a3 = s + 3.f, d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    for (int i = 0; i < n; ++i) {
        REP16(asm volatile("v_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                           "v_mov_b32_dpp %2, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                           : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3;
}

template <typename K>
double run(K kernel, float *out, int blocks, int n)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), 0, 0, out, 16, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), 0, 0, out, n, 1.0f);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main()
{
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    int clock_khz = 0;
    CHECK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0));
    printf("compute units %d, clock attribute %d kHz\n", cus, clock_khz);
    float *out = nullptr;
    CHECK(hipMalloc((void **)&out, (size_t)cus * 4 * 8 * 64 * sizeof(float)));
    const int n = 4000;  // x 64 instruction slots
    printf("%-34s %8s %8s %8s %8s   (ns per 64-lane instruction slot and wavefront)\n", "kernel \\ waves per SIMD", "1", "2", "3", "4");
    struct { const char *name; int which; } rows[] = {
        {"f32 fma, dependent", 0}, {"f32 fma, 4 independent", 1}, {"pk fma dependent + s_nop 0", 2},
        {"pk fma, 4 independent", 3}, {"pk fma, 2 independent", 4}, {"pk fma 4 indep + s_add each", 5},
        {"pk fma 4 indep + s_nop 0 each", 6}, {"v_mov dpp wave_shr, 4 independent", 7}};
    for (auto &r : rows) {
        printf("%-34s", r.name); fflush(stdout);
        for (int w = 1; w <= 4; ++w) {
            const int blocks = cus * 4 * w;
            double ms = 0;
            switch (r.which) {
                case 0: ms = run(k_f32_dep, out, blocks, n); break;
                case 1: ms = run(k_f32_indep, out, blocks, n); break;
                case 2: ms = run(k_pk_dep_nop, out, blocks, n); break;
                case 3: ms = run(k_pk_indep, out, blocks, n); break;
                case 4: ms = run(k_pk_indep2, out, blocks, n); break;
                case 5: ms = run(k_pk_indep_salu, out, blocks, n); break;
                case 6: ms = run(k_pk_indep_nop, out, blocks, n); break;
                case 7: ms = run(k_dpp_mov, out, blocks, n); break;
            }
            // time per VALU instruction of one wavefront
            printf(" %8.3f", ms * 1e6 / ((double)n * 64)); fflush(stdout);
        }
        printf("\n");
    }
    CHECK(hipFree(out));
    return 0;
}
