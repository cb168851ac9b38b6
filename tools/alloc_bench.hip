// hipMalloc / hipFree cost by size (planner design input).  hipcc --offload-arch=gfx950 -O2 tools/alloc_bench.hip -o /tmp/alloc_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    void *warm;
    hipMalloc(&warm, 1 << 20);
    for (size_t mb : {64, 512, 4096, 16384}) {
        for (int rep = 0; rep < 2; rep++) {
            void *p = nullptr;
            double t0 = now();
            hipError_t e = hipMalloc(&p, mb << 20);
            double t1 = now();
            hipMemset(p, 0, mb << 20);
            hipDeviceSynchronize();
            double t2 = now();
            hipFree(p);
            double t3 = now();
            std::printf("%6zu MiB: hipMalloc %8.2f ms  memset %8.2f ms  hipFree %8.2f ms (%d)\n", mb, t1 - t0, t2 - t1, t3 - t2, (int)e);
        }
    }
    return 0;
}
