// xcd_atomic_bench.hip - what bounds the EXPANSION of a sweep pass at BASELINE size (C4's first sweep pass: 61 M (changed source,
// reader row) pairs -> bits in a 16 MB touch bitmap, 3.03 ms = 20 G pairs/s; profiles/r05c_C4_kernel_stats.csv) and whether an
// XCD-sliced form of it is worth building.  N reader-row ids (uniform over `rows`, read coalesced like the reader lists) each set
// one bit of a rows / 8-byte bitmap.  Variants:
//   agent+pretest   today's touch_set(): device-coherent pre-test load, then atomicOr at agent scope if the bit is not set yet
//   agent           atomicOr at agent scope for every pair (no pre-test)
//   sliced-wg       every workgroup reads the XCC it runs on (HW_REG_XCC_ID), pulls chunks of the pair list from ITS XCC's queue and
//                   sets only the bits of rows in that XCC's eighth of the bitmap (2 MB: L2-resident), with WORKGROUP-scope atomics
//                   (performed in the XCD's L2; no other XCD touches those words inside the launch).  Every XCC walks the whole list.
//   sliced-agent    the same slicing with agent-scope atomics (separates what the slicing buys from what the scope buys)
//   unsliced-wg     workgroup-scope atomics with NO slicing - WRONG by construction (two L2s hold the same word): shows that the
//                   hazard is real on this machine (bits get lost), i.e. that the sliced form's correctness rests on the slicing
//   lds-owner       no global atomics at all: the bitmap is cut into 256 Ki-row pieces (32 KB of bits), a workgroup stages ONE piece in
//                   LDS, walks the whole list, ORs the bits of its piece in LDS and stores the piece once (upper bound of the
//                   re-read cost of an owner-computes form without a partition pass)
// The result bitmap of every variant is compared with the host's.  usage: xcd_atomic_bench.bin [rows [pairs]]
// Build: hipcc --offload-arch=gfx950 -O3 tools/xcd_atomic_bench.hip -o tools/xcd_atomic_bench.bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                \
    do {                                                                     \
        hipError_t e_ = (x);                                                 \
        if (e_ != hipSuccess) {                                              \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));              \
            std::exit(1);                                                    \
        }                                                                    \
    } while (0)

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 7u;
}

__global__ __launch_bounds__(256) void k_agent(const uint32_t *idx, uint64_t n, uint32_t *bits, int pretest)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t r = idx[i], bit = 1u << (r & 31u);
        if (pretest && (__hip_atomic_load(&bits[r >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) continue;
        atomicOr(&bits[r >> 5], bit);
    }
}

__global__ __launch_bounds__(256) void k_unsliced_wg(const uint32_t *idx, uint64_t n, uint32_t *bits)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t r = idx[i], bit = 1u << (r & 31u);
        __hip_atomic_fetch_or(&bits[r >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

constexpr uint32_t kChunk = 256 * 16; // pairs per queue pull
// heads[x] = next chunk of XCC x's walk over the list; seen[x] = workgroups that found themselves on XCC x
template <bool WG_SCOPE>
__global__ __launch_bounds__(256) void k_sliced(const uint32_t *idx, uint64_t n, uint32_t *bits, uint64_t rows, unsigned int *heads, unsigned int *seen)
{
    __shared__ uint32_t s_chunk;
    const uint32_t x = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&seen[x], 1u);
    const uint64_t per = (rows + 7) / 8 + 31 & ~31ull; // rows per XCC slice, whole bitmap words
    const uint64_t lo = (uint64_t)x * per, hi = lo + per;
    const uint64_t nchunks = (n + kChunk - 1) / kChunk;
    for (;;) {
        if (threadIdx.x == 0) s_chunk = atomicAdd(&heads[x], 1u);
        __syncthreads();
        const uint64_t c = s_chunk;
        __syncthreads();
        if (c >= nchunks) break;
        const uint64_t base = c * kChunk;
        uint32_t r[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
            r[k] = i < n ? idx[i] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (r[k] < lo || r[k] >= hi) continue;
            const uint32_t bit = 1u << (r[k] & 31u);
            uint32_t *w = &bits[r[k] >> 5];
            if (WG_SCOPE) {
                if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & bit) continue;
                __hip_atomic_fetch_or(w, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) continue;
                atomicOr(w, bit);
            }
        }
    }
}

constexpr uint32_t kPieceRows = 1u << 18; // 32 KB of bits
__global__ __launch_bounds__(1024) void k_lds_owner(const uint32_t *idx, uint64_t n, uint32_t *bits, uint64_t rows)
{
    __shared__ uint32_t s_bits[kPieceRows / 32];
    const uint64_t pieces = (rows + kPieceRows - 1) / kPieceRows;
    for (uint64_t p = blockIdx.x; p < pieces; p += gridDim.x) {
        for (uint32_t i = threadIdx.x; i < kPieceRows / 32; i += 1024) s_bits[i] = 0;
        __syncthreads();
        const uint32_t lo = (uint32_t)(p * kPieceRows);
        for (uint64_t base = 0; base < n; base += 1024 * 8) {
            uint32_t r[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint64_t i = base + (uint64_t)k * 1024 + threadIdx.x;
                r[k] = i < n ? idx[i] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t d = r[k] - lo;
                if (d < kPieceRows) atomicOr(&s_bits[d >> 5], 1u << (d & 31u));
            }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < kPieceRows / 32; i += 1024) {
            const uint64_t w = (uint64_t)lo / 32 + i;
            if (w < (rows + 31) / 32) bits[w] = s_bits[i];
        }
        __syncthreads();
    }
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
    const uint64_t rows = argc > 1 ? (uint64_t)std::atoll(argv[1]) : 130000000ull; // C4: n_pad + virtual rows
    const uint64_t n = argc > 2 ? (uint64_t)std::atoll(argv[2]) : 61000000ull;      // C4's first sweep pass: A = 2.9 % of the edges
    const uint64_t words = (rows + 31) / 32 + 64;
    std::vector<uint32_t> idx(n), want(words, 0), got(words);
    uint64_t s = 11;
    for (uint64_t i = 0; i < n; i++) {
        idx[i] = (uint32_t)(sm(s) % rows);
        want[idx[i] >> 5] |= 1u << (idx[i] & 31u);
    }
    uint32_t *d_idx, *d_bits;
    unsigned int *d_q;
    CK(hipMalloc(&d_idx, n * 4));
    CK(hipMalloc(&d_bits, words * 4));
    CK(hipMalloc(&d_q, 64));
    CK(hipMemcpy(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    std::printf("%llu pairs -> bits of %llu rows (bitmap %.1f MB), %d CUs\n", (unsigned long long)n, (unsigned long long)rows, rows / 8e6, cus);
    struct V {
        const char *name;
        int kind, wg_per_cu;
    } variants[] = {{"agent + pre-test (today), 4 wg/CU", 0, 4}, {"agent + pre-test, 8 wg/CU", 0, 8}, {"agent, no pre-test, 4 wg/CU", 1, 4},
                    {"sliced by XCC, workgroup-scope, 4 wg/CU", 2, 4}, {"sliced by XCC, workgroup-scope, 8 wg/CU", 2, 8}, {"sliced by XCC, agent-scope, 4 wg/CU", 3, 4},
                    {"UNSLICED workgroup-scope (expected wrong)", 4, 4}, {"LDS owner pieces, no global atomics, 1 wg/CU", 5, 1}};
    for (const V &v : variants) {
        float best = 1e9f;
        unsigned int seen[16] = {0};
        uint64_t diff = 0;
        for (int it = 0; it < 3; it++) {
            CK(hipMemset(d_bits, 0, words * 4));
            CK(hipMemset(d_q, 0, 64));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            const dim3 grid((unsigned)(cus * v.wg_per_cu));
            if (v.kind == 0) hipLaunchKernelGGL(k_agent, grid, dim3(256), 0, 0, d_idx, n, d_bits, 1);
            else if (v.kind == 1) hipLaunchKernelGGL(k_agent, grid, dim3(256), 0, 0, d_idx, n, d_bits, 0);
            else if (v.kind == 2) hipLaunchKernelGGL(k_sliced<true>, grid, dim3(256), 0, 0, d_idx, n, d_bits, rows, d_q, d_q + 8);
            else if (v.kind == 3) hipLaunchKernelGGL(k_sliced<false>, grid, dim3(256), 0, 0, d_idx, n, d_bits, rows, d_q, d_q + 8);
            else if (v.kind == 4) hipLaunchKernelGGL(k_unsliced_wg, grid, dim3(256), 0, 0, d_idx, n, d_bits);
            else hipLaunchKernelGGL(k_lds_owner, grid, dim3(1024), 0, 0, d_idx, n, d_bits, rows);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            CK(hipGetLastError());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            best = ms < best ? ms : best;
            CK(hipMemcpy(got.data(), d_bits, words * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(seen, d_q, 64, hipMemcpyDeviceToHost));
            diff = 0;
            for (uint64_t w = 0; w < words; w++) diff += (uint64_t)__builtin_popcount(got[w] ^ want[w]);
        }
        std::printf("  %-48s %7.3f ms  %6.1f G pairs/s  %s", v.name, best, (double)n / best / 1e6, diff == 0 ? "(bitmap ok)" : "(BITMAP DIFFERS:");
        if (diff) std::printf(" %llu bits)", (unsigned long long)diff);
        if (v.kind == 2 || v.kind == 3) {
            std::printf("  workgroups per XCC:");
            for (int x = 0; x < 8; x++) std::printf(" %u", seen[8 + x]);
        }
        std::printf("\n");
    }
    return 0;
}
