#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now().time_since_epoch()).count(); }
int main() {
    hipSetDevice(0); hipFree(nullptr);
    const size_t b = (size_t)9 * 16384 * 16384 * 4;
    for (int rep = 0; rep < 2; ++rep) {
        void *p, *q, *r;
        double t0 = now(); hipMalloc(&p, b); double t1 = now(); hipMalloc(&q, b); double t2 = now();
        hipMemsetAsync(p, 0, b, 0); hipDeviceSynchronize(); double t3 = now();
        hipMemsetAsync(p, 0, b, 0); hipDeviceSynchronize(); double t4 = now();
        hipFree(p); hipFree(q); double t5 = now();
        hipMalloc(&r, 2 * b); double t6 = now(); hipFree(r); double t7 = now();
        printf("rep %d: malloc 9.66GB %.2f ms, second %.2f ms, first touch (memset) %.2f ms, second memset %.2f ms, 2 frees %.2f ms, one 19.3GB malloc %.2f ms, free %.2f ms\n", rep, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6);
    }
    return 0;
}
