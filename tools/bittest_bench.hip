// bittest_bench.hip - what bounds the bitmap pass (~209 M changed-bit tests per pass at C3, L2 hits at ~170 G/s):
// is a test against an LDS-resident part of the bitmap measurably cheaper than the L2 hit it replaces?  Decides next-step 3 of
// DESIGN.md §8 BEFORE any kernel is rebuilt around it.
//
// The stream: N source indices (4 B each, read coalesced like the index lists of the work rows), hot-skewed like the device order
// makes them - a share `hot` of them below `hot_rows` (C3: 80 % below 512 Ki), the rest uniform over `rows`; every index is tested
// against a bitmap of rows/8 bytes with `density` of its bits set; the kernel counts the hits (so nothing is optimised away) and
// does nothing else.  Variants:
//   global      every test is a 4-byte load from the bitmap in global memory (what frontier_kernel does today)
//   lds<B>      the first `hot_rows` bits are staged into LDS once per workgroup of B lanes (64 KiB for 512 Ki rows); tests below
//               that bound read LDS, the others global memory.  B = 256 caps the CU at 2 workgroups = 2 waves per SIMD (the
//               round-3 experiment that lost), B = 1024 keeps 8 waves per SIMD with 2 workgroups per CU
// Output: G tests/s per variant.  usage: bittest_bench.bin [rows [hot_rows [hot_share [density [tests]]]]]  Build: hipcc --offload-arch=gfx950 -O3 tools/bittest_bench.hip -o tools/bittest_bench.bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                \
    do {                                                                     \
        hipError_t e_ = (x);                                                 \
        if (e_ != hipSuccess) {                                              \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));              \
            std::exit(1);                                                    \
        }                                                                    \
    } while (0)

constexpr int kPerLane = 8; // indices per lane and iteration (the batched frontier kernel keeps 16 in flight per lane)

__global__ __launch_bounds__(256) void test_global(const uint32_t *idx, uint64_t n, const uint32_t *bits, unsigned long long *hits)
{
    unsigned long long h = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * kPerLane;
    for (uint64_t base = ((uint64_t)blockIdx.x * blockDim.x) * kPerLane; base < n; base += stride) {
        uint32_t id[kPerLane], w[kPerLane];
#pragma unroll
        for (int k = 0; k < kPerLane; k++) {
            const uint64_t i = base + (uint64_t)k * blockDim.x + threadIdx.x;
            id[k] = i < n ? idx[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < kPerLane; k++) w[k] = bits[id[k] >> 5];
#pragma unroll
        for (int k = 0; k < kPerLane; k++) h += (w[k] >> (id[k] & 31u)) & 1u;
    }
    for (int off = 32; off > 0; off >>= 1) h += __shfl_down(h, off);
    if ((threadIdx.x & 63) == 0 && h) atomicAdd(hits, h);
}

template <int B>
__global__ __launch_bounds__(B) void test_lds(const uint32_t *idx, uint64_t n, const uint32_t *bits, uint32_t hot_rows, unsigned long long *hits)
{
    extern __shared__ uint32_t s_bits[]; // hot_rows / 32 words
    for (uint32_t i = threadIdx.x; i < hot_rows / 32; i += B) s_bits[i] = bits[i];
    __syncthreads();
    unsigned long long h = 0;
    const uint64_t stride = (uint64_t)gridDim.x * B * kPerLane;
    for (uint64_t base = ((uint64_t)blockIdx.x * B) * kPerLane; base < n; base += stride) {
        uint32_t id[kPerLane], w[kPerLane];
#pragma unroll
        for (int k = 0; k < kPerLane; k++) {
            const uint64_t i = base + (uint64_t)k * B + threadIdx.x;
            id[k] = i < n ? idx[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < kPerLane; k++) w[k] = id[k] < hot_rows ? s_bits[id[k] >> 5] : bits[id[k] >> 5];
#pragma unroll
        for (int k = 0; k < kPerLane; k++) h += (w[k] >> (id[k] & 31u)) & 1u;
    }
    for (int off = 32; off > 0; off >>= 1) h += __shfl_down(h, off);
    if ((threadIdx.x & 63) == 0 && h) atomicAdd(hits, h);
}

static uint64_t sm(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
    const uint32_t rows = argc > 1 ? (uint32_t)std::atol(argv[1]) : (1u << 23); // 8 Mi rows = 1 MiB of bitmap
    const uint32_t hot_rows = argc > 2 ? (uint32_t)std::atol(argv[2]) : (1u << 19); // 64 KiB of it
    const double hot = argc > 3 ? std::atof(argv[3]) : 0.8, density = argc > 4 ? std::atof(argv[4]) : 0.16;
    const uint64_t n = argc > 5 ? (uint64_t)std::atoll(argv[5]) : (1ull << 28); // 256 Mi tests (C3: 209 M per bitmap pass)
    std::vector<uint32_t> idx(n), bits(rows / 32);
    uint64_t s = 7;
    for (uint64_t i = 0; i < n; i++) {
        const uint64_t r = sm(s);
        idx[i] = ((double)(r >> 40) / (double)(1 << 24) < hot) ? (uint32_t)(r % hot_rows) : (uint32_t)(r % rows);
    }
    unsigned long long want = 0;
    for (auto &w : bits) {
        w = 0;
        for (int b = 0; b < 32; b++)
            if ((double)(sm(s) >> 40) / (double)(1 << 24) < density) w |= 1u << b;
    }
    for (uint64_t i = 0; i < n; i++) want += (bits[idx[i] >> 5] >> (idx[i] & 31u)) & 1u;
    uint32_t *d_idx, *d_bits;
    unsigned long long *d_hits;
    CK(hipMalloc(&d_idx, n * 4));
    CK(hipMalloc(&d_bits, rows / 8));
    CK(hipMalloc(&d_hits, 8));
    CK(hipMemcpy(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bits, bits.data(), rows / 8, hipMemcpyHostToDevice));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const size_t lds = (size_t)hot_rows / 8;
    CK(hipFuncSetAttribute((const void *)test_lds<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)test_lds<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::printf("%llu tests, bitmap %u KiB (LDS part %zu KiB), %.0f %% of the indices below the LDS bound, density %.2f, %d CUs\n", (unsigned long long)n, rows / 8192, lds / 1024,
                hot * 100, density, cus);
    struct V {
        const char *name;
        int kind, wg_per_cu;
    } variants[] = {{"global, 5 workgroups of 256 per CU (today)", 0, 5}, {"global, 8 per CU", 0, 8}, {"lds<256>, 2 per CU", 1, 2}, {"lds<1024>, 2 per CU", 2, 2},
                    {"lds<1024>, 1 per CU", 2, 1}};
    for (const V &v : variants) {
        float best = 1e9f;
        unsigned long long got = 0;
        for (int it = 0; it < 4; it++) {
            CK(hipMemset(d_hits, 0, 8));
            CK(hipEventRecord(a));
            const dim3 grid((unsigned)(cus * v.wg_per_cu));
            if (v.kind == 0) hipLaunchKernelGGL(test_global, grid, dim3(256), 0, 0, d_idx, n, d_bits, d_hits);
            else if (v.kind == 1) hipLaunchKernelGGL(test_lds<256>, grid, dim3(256), lds, 0, d_idx, n, d_bits, hot_rows, d_hits);
            else hipLaunchKernelGGL(test_lds<1024>, grid, dim3(1024), lds, 0, d_idx, n, d_bits, hot_rows, d_hits);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            CK(hipGetLastError());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            best = ms < best ? ms : best;
            CK(hipMemcpy(&got, d_hits, 8, hipMemcpyDeviceToHost));
        }
        std::printf("  %-44s %7.3f ms  %7.1f G tests/s  %s\n", v.name, best, (double)n / best / 1e6, got == want ? "(count ok)" : "(COUNT WRONG)");
    }
    return 0;
}
