// What a grid-wide barrier costs inside one launch on gfx950, next to the gap between two dependent
// launches: the price list for a persistent small-lattice kernel (many blocks of steps per launch).
// Every workgroup is one wavefront, all co-resident (grid <= 2 per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2);} } while (0)

// sense-reversing counter barrier; FENCE: 0 none (relaxed atomics only), 1 agent-scope release before /
// acquire after (what makes the stores of all XCDs visible to the loads of all XCDs)
template <int FENCE>
__global__ void k_barriers(unsigned *count, unsigned *gen, float *data, int n, unsigned nblocks)
{
    float v = data[blockIdx.x * 64 + threadIdx.x];
    unsigned my_gen = 0;
    for (int i = 0; i < n; ++i) {
        v = v * 1.0001f + 1.0f;
        data[((blockIdx.x + i) % nblocks) * 64 + threadIdx.x] = v;     // something to publish
        if (FENCE) __atomic_thread_fence(__ATOMIC_RELEASE);            // agent scope is HIP's default for __atomic_thread_fence? use builtin below
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (threadIdx.x == 0) {
            const unsigned arrived = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            if (arrived == nblocks * (my_gen + 1)) {
                __hip_atomic_store(gen, my_gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= my_gen) __builtin_amdgcn_s_sleep(1);
            }
        }
        ++my_gen;
        __builtin_amdgcn_s_barrier();
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        v += data[((blockIdx.x + i + 1) % nblocks) * 64 + threadIdx.x];   // read a neighbour's value
    }
    data[blockIdx.x * 64 + threadIdx.x] = v;
}

__global__ void k_step(float *data, unsigned nblocks, int i)
{
    float v = data[blockIdx.x * 64 + threadIdx.x];
    v = v * 1.0001f + 1.0f + data[((blockIdx.x + i + 1) % nblocks) * 64 + threadIdx.x];
    data[blockIdx.x * 64 + threadIdx.x] = v;
}

int main()
{
    const int n = 2000;
    for (unsigned nblocks : {256u, 1024u, 2048u}) {
        unsigned *count, *gen; float *data;
        CK(hipMalloc(&count, 4)); CK(hipMalloc(&gen, 4)); CK(hipMalloc(&data, (size_t)nblocks * 64 * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms[3] = {0, 0, 0};
        for (int variant = 0; variant < 3; ++variant) {
            CK(hipMemset(count, 0, 4)); CK(hipMemset(gen, 0, 4)); CK(hipMemset(data, 0, (size_t)nblocks * 64 * 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            if (variant == 0) hipLaunchKernelGGL(k_barriers<0>, dim3(nblocks), dim3(64), 0, 0, count, gen, data, n, nblocks);
            else if (variant == 1) hipLaunchKernelGGL(k_barriers<1>, dim3(nblocks), dim3(64), 0, 0, count, gen, data, n, nblocks);
            else for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_step, dim3(nblocks), dim3(64), 0, 0, data, nblocks, i);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[variant], e0, e1));
        }
        printf("%4u wavefronts: grid barrier, relaxed atomics only %6.2f us | with agent-scope release + acquire %6.2f us | one tiny kernel per step (launch to launch) %6.2f us\n",
               nblocks, ms[0] * 1e3 / n, ms[1] * 1e3 / n, ms[2] * 1e3 / n);
        fflush(stdout);
        CK(hipFree(count)); CK(hipFree(gen)); CK(hipFree(data));
    }
    return 0;
}
