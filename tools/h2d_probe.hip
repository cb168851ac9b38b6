// h2d_probe.hip - what the host link gives hb_append_edges: pinned H2D by hipMemcpyAsync at several transfer sizes, one and two
// streams, against a kernel that reads the pinned host buffer directly (zero copy), and pageable memory for comparison.
//   hipcc --offload-arch=gfx950 -O2 tools/h2d_probe.hip -o tools/h2d_probe.bin
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));               \
            return 1;                                                         \
        }                                                                     \
    } while (0)

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// every thread reads 16 bytes of host memory per step and keeps a checksum (so the loads are not dropped)
__global__ __launch_bounds__(256) void read_host_kernel(const uint4 *src, size_t n16, uint4 *dst)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

int main()
{
    const size_t cap = 1ull << 30;
    void *h_pinned = nullptr, *h_page = std::malloc(cap), *d = nullptr, *d2 = nullptr;
    CK(hipHostMalloc(&h_pinned, cap));
    std::memset(h_pinned, 1, cap);
    std::memset(h_page, 2, cap);
    CK(hipMalloc(&d, cap));
    CK(hipMalloc(&d2, cap));
    hipStream_t s[2];
    CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    for (size_t sz : {(size_t)16 << 20, (size_t)64 << 20, (size_t)160 << 20, (size_t)640 << 20, (size_t)1 << 30}) {
        const int reps = (int)((4ull << 30) / sz);
        CK(hipMemcpyAsync(d, h_pinned, sz, hipMemcpyHostToDevice, s[0]));
        CK(hipStreamSynchronize(s[0]));
        double t0 = now();
        for (int r = 0; r < reps; r++) CK(hipMemcpyAsync(d, h_pinned, sz, hipMemcpyHostToDevice, s[0]));
        CK(hipStreamSynchronize(s[0]));
        double t1 = now();
        std::printf("pinned H2D hipMemcpyAsync %5zu MiB x %3d, 1 stream : %6.2f GB/s\n", sz >> 20, reps, (double)sz * reps / (t1 - t0) / 1e9);
        t0 = now();
        for (int r = 0; r < reps; r++) CK(hipMemcpyAsync(r & 1 ? d2 : d, (char *)h_pinned + ((r & 1) ? sz % (cap - sz + 1) : 0), sz, hipMemcpyHostToDevice, s[r & 1]));
        CK(hipStreamSynchronize(s[0]));
        CK(hipStreamSynchronize(s[1]));
        t1 = now();
        std::printf("pinned H2D hipMemcpyAsync %5zu MiB x %3d, 2 streams: %6.2f GB/s\n", sz >> 20, reps, (double)sz * reps / (t1 - t0) / 1e9);
    }
    {
        const size_t sz = (size_t)640 << 20;
        for (unsigned grid : {256u, 1024u, 4096u}) {
            hipLaunchKernelGGL(read_host_kernel, dim3(grid), dim3(256), 0, s[0], (const uint4 *)h_pinned, sz / 16, (uint4 *)d);
            CK(hipStreamSynchronize(s[0]));
            const double t0 = now();
            for (int r = 0; r < 4; r++) hipLaunchKernelGGL(read_host_kernel, dim3(grid), dim3(256), 0, s[0], (const uint4 *)h_pinned, sz / 16, (uint4 *)d);
            CK(hipStreamSynchronize(s[0]));
            const double t1 = now();
            std::printf("kernel reading pinned host memory (zero copy), %4u blocks    : %6.2f GB/s\n", grid, (double)sz * 4 / (t1 - t0) / 1e9);
        }
    }
    {
        const size_t sz = (size_t)640 << 20;
        const double t0 = now();
        for (int r = 0; r < 3; r++) CK(hipMemcpyAsync(d, h_page, sz, hipMemcpyHostToDevice, s[0]));
        CK(hipStreamSynchronize(s[0]));
        const double t1 = now();
        std::printf("PAGEABLE H2D hipMemcpyAsync 640 MiB x 3                        : %6.2f GB/s\n", (double)sz * 3 / (t1 - t0) / 1e9);
        // registering the caller's pageable buffer first (what a shim could do once for a reused batch buffer)
        double r0 = now();
        CK(hipHostRegister(h_page, cap, hipHostRegisterDefault));
        double r1 = now();
        const double t2 = now();
        for (int r = 0; r < 3; r++) CK(hipMemcpyAsync(d, h_page, sz, hipMemcpyHostToDevice, s[0]));
        CK(hipStreamSynchronize(s[0]));
        const double t3 = now();
        std::printf("hipHostRegister(1 GiB) %.3f s; then H2D 640 MiB x 3                : %6.2f GB/s\n", r1 - r0, (double)sz * 3 / (t3 - t2) / 1e9);
        CK(hipHostUnregister(h_page));
    }
    {
        // D2H for the result download / store emission
        const size_t sz = (size_t)640 << 20;
        const double t0 = now();
        for (int r = 0; r < 4; r++) CK(hipMemcpyAsync(h_pinned, d, sz, hipMemcpyDeviceToHost, s[0]));
        CK(hipStreamSynchronize(s[0]));
        const double t1 = now();
        std::printf("pinned D2H hipMemcpyAsync 640 MiB x 4                          : %6.2f GB/s\n", (double)sz * 4 / (t1 - t0) / 1e9);
    }
    return 0;
}
