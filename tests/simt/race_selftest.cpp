// tests/simt/race_selftest.cpp - TEST INFRASTRUCTURE: is the cross-workgroup race detector awake?  Built with -fsanitize=thread and run
// with HB_SIMT_THREADS > 1 (workgroups of a launch on several host threads).  "racy": every workgroup adds to one counter with a
// plain read-modify-write - ThreadSanitizer must report it; no argument: the same with atomicAdd - it must stay silent.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

__global__ void count_kernel(unsigned *counter, int racy)
{
    const uint64_t any = __ballot(1);
    if ((threadIdx.x & 63) == 0) {
        if (racy) *counter = *counter + (unsigned)__popcll(any);
        else atomicAdd(counter, (unsigned)__popcll(any));
    }
}

int main(int argc, char **argv)
{
    const int racy = argc > 1 && std::strcmp(argv[1], "racy") == 0;
    unsigned *d = nullptr, h = 0;
    if (hipMalloc(&d, sizeof(unsigned)) != hipSuccess) return 2;
    hipMemset(d, 0, sizeof(unsigned));
    for (int it = 0; it < 20; it++) hipLaunchKernelGGL(count_kernel, dim3(64), dim3(256), 0, nullptr, d, racy);
    hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    std::printf("%s %u\n", racy ? "racy" : "atomic", h);
    return 0;
}
