// tests/simt/asan_selftest.cpp - TEST INFRASTRUCTURE: is the memory checker of the interpreted build awake?  A kernel whose last lane
// reads ONE element behind a hipMalloc'ed buffer (argv[1] = "over") or stays inside (no argument): AddressSanitizer must abort
// the first and let the second print "clean".
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

__global__ void sum_kernel(const uint32_t *in, uint32_t n, uint32_t *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t v = i < n ? in[i] : 0u;
    const uint64_t any = __ballot(v != 0);
    if ((threadIdx.x & 63) == 0 && any) atomicAdd(out, (uint32_t)__popcll(any));
}

int main(int argc, char **argv)
{
    const bool over = argc > 1 && std::strcmp(argv[1], "over") == 0;
    const uint32_t n = 1000;
    uint32_t *d_in = nullptr, *d_out = nullptr, h = 0;
    if (hipMalloc(&d_in, n * sizeof(uint32_t)) != hipSuccess || hipMalloc(&d_out, sizeof(uint32_t)) != hipSuccess) return 2;
    hipMemset(d_in, 1, n * sizeof(uint32_t));
    hipMemset(d_out, 0, sizeof(uint32_t));
    hipLaunchKernelGGL(sum_kernel, dim3(4), dim3(256), 0, nullptr, (const uint32_t *)d_in, over ? n + 1 : n, d_out);
    hipMemcpy(&h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    std::printf("%s %u\n", "clean", h);
    return 0;
}
