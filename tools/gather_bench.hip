// gather_bench.hip - ceiling of the access pattern the HyperBall pass is made of: one quad
// (4 lanes x 16 B) gathers one 64-byte counter, byte-max accumulates, 64 gathers per row.
// Measures GB/s of gathered bytes for source indices drawn uniformly from a region of R
// counters (R*64 B = 4 MiB one L2 ... 8 GiB HBM) and for an exactly-once permutation (known
// byte count: calibrates rocprofv3 FETCH_SIZE for 64-byte gathers on gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o gpurun_out/gather_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pkmax(uint32_t a, uint32_t b)
{
    us2 x = __builtin_bit_cast(us2, a), y = __builtin_bit_cast(us2, b);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(x, y));
}
template <int J> __device__ __forceinline__ uint32_t qb(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, J * 0x55, 0xF, 0xF, true);
}

template <int MODE> __device__ __forceinline__ uint4 ld(const uint4 *p)
{
    if (MODE == 1) {
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        v4u v = __builtin_nontemporal_load((const v4u *)p);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *p;
}

template <int UNROLL, int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const uint32_t *idx, const uint4 *regs, uint4 *out, uint64_t rows, int deg)
{
    const int lane = threadIdx.x & 63, q = lane & 3;
    const uint64_t ntiles = (rows + 63) / 64;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t row = tile * 64 + (threadIdx.x >> 2);
        if (row >= rows) continue;
        uint32_t e[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0};
        const uint32_t *p = idx + row * deg;
        for (int k = 0; k < deg; k += 4 * UNROLL) {
            uint32_t id[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) id[u] = p[k + 4 * u + q];
            uint4 r[UNROLL][4];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                if (MODE == 5) { // round 5: HALF a counter per gather (32 B per quad, 8 B per lane): is the miss path a request limit
                                 // whatever the size (then a 32-B compressed counter buys nothing), or does 32 B go faster?
                    const uint2 *h = (const uint2 *)regs;
                    const uint2 v0 = h[(uint64_t)qb<0>(id[u]) * 8 + q], v1 = h[(uint64_t)qb<1>(id[u]) * 8 + q];
                    const uint2 v2 = h[(uint64_t)qb<2>(id[u]) * 8 + q], v3 = h[(uint64_t)qb<3>(id[u]) * 8 + q];
                    r[u][0] = make_uint4(v0.x, v0.y, 0, 0);
                    r[u][1] = make_uint4(v1.x, v1.y, 0, 0);
                    r[u][2] = make_uint4(v2.x, v2.y, 0, 0);
                    r[u][3] = make_uint4(v3.x, v3.y, 0, 0);
                } else if (MODE == 6) { // round 5: NEIGHBOURING QUADS of a wave gather the two halves of one 128-B line in the same
                                        // instruction (quad 2k takes the even counter of quad 2k's index, quad 2k+1 its line mate):
                                        // does the texture addresser merge them into one request like the same-quad pair of MODE 2?
                    const uint32_t odd = (lane >> 2) & 1u;
                    uint32_t a[4] = {qb<0>(id[u]), qb<1>(id[u]), qb<2>(id[u]), qb<3>(id[u])};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t mine = a[j], other = (uint32_t)__shfl_xor((int)mine, 4); // the neighbouring quad's index
                        const uint32_t base = (odd ? other : mine) & ~1u;
                        r[u][j] = regs[(uint64_t)(base | odd) * 4 + q];
                    }
                } else if (MODE == 2) { // two adjacent counters (one 128-B line) per index: idx pairs (2k, 2k+1)
                    const uint32_t a0 = qb<0>(id[u]) & ~1u, a1 = qb<2>(id[u]) & ~1u;
                    r[u][0] = regs[(uint64_t)a0 * 4 + q];
                    r[u][1] = regs[(uint64_t)a0 * 4 + 4 + q];
                    r[u][2] = regs[(uint64_t)a1 * 4 + q];
                    r[u][3] = regs[(uint64_t)a1 * 4 + 4 + q];
                } else {
                    r[u][0] = ld<MODE>(regs + (uint64_t)qb<0>(id[u]) * 4 + q);
                    r[u][1] = ld<MODE>(regs + (uint64_t)qb<1>(id[u]) * 4 + q);
                    r[u][2] = ld<MODE>(regs + (uint64_t)qb<2>(id[u]) * 4 + q);
                    r[u][3] = ld<MODE>(regs + (uint64_t)qb<3>(id[u]) * 4 + q);
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t w[4] = {r[u][j].x, r[u][j].y, r[u][j].z, r[u][j].w};
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        e[c] = pkmax(e[c], w[c] & 0x00FF00FFu);
                        o[c] = pkmax(o[c], w[c] & 0xFF00FF00u);
                    }
                }
        }
        out[row * 4 + q] = make_uint4(e[0] | o[0], e[1] | o[1], e[2] | o[2], e[3] | o[3]);
    }
}

__global__ void stream_kernel(const uint4 *in, uint4 *out, uint64_t n4)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * 256) out[i] = in[i];
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
    const uint64_t total = 1ull << 27;       // counters in the big array: 8 GiB
    const uint64_t rows = 1ull << 21;        // 2 Mi rows x 64 gathers = 128 Mi gathers = 8 GiB gathered
    const int deg = 64;
    const char *only = argc > 1 ? argv[1] : "";
    uint4 *regs, *out;
    uint32_t *d_idx;
    CK(hipMalloc(&regs, total * 64));
    CK(hipMalloc(&out, rows * 64));
    CK(hipMalloc(&d_idx, rows * deg * 4));
    CK(hipMemset(regs, 1, total * 64));
    std::vector<uint32_t> idx(rows * deg);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    struct Case { const char *name; uint64_t region; int perm; };
    Case cases[] = {{"L2_4MiB", 1ull << 16, 0},   {"L2x8_32MiB", 1ull << 19, 0}, {"MALL_128MiB", 1ull << 21, 0},
                    {"MALL_256MiB", 1ull << 22, 0}, {"HBM_512MiB", 1ull << 23, 0}, {"HBM_8GiB", 1ull << 27, 0},
                    {"perm_8GiB_once", 1ull << 27, 1}, {"seq_8GiB_once", 1ull << 27, 2},
                    // round 3: does address ORDER or a moving WINDOW lift the ~45 G/s of random 64-byte gathers?
                    // sorted_rows: the 64 sources of every row ascending (what sorting the sources inside a chunk would give);
                    // window_*: rows processed at the same time draw their sources from one window of that many counters that
                    // slides over the 8 GiB (what a band-major / locality-aware order of cold sources would give)
                    {"HBM_8GiB_sorted_rows", 1ull << 27, 3},
                    {"window_4MiB_in_8GiB", 1ull << 16, 4}, {"window_32MiB_in_8GiB", 1ull << 19, 4}, {"window_256MiB_in_8GiB", 1ull << 22, 4},
                    {"window_1GiB_in_8GiB", 1ull << 24, 4}};
    for (const Case &c : cases) {
        if (only[0] && !strstr(only, c.name)) continue; // argv[1]: comma-separated case names; argv[2]: first variant
        uint64_t s = 42;
        if (c.perm == 1) {
            // exactly-once: a random permutation of all 2^27 blocks (multiplicative + xor shuffle)
            for (uint64_t i = 0; i < rows * deg; i++) idx[i] = (uint32_t)(((i * 0x9E3779B1ull) ^ 0x5A5A5A5ull) & (total - 1));
        } else if (c.perm == 2) {
            for (uint64_t i = 0; i < rows * deg; i++) idx[i] = (uint32_t)i;
        } else if (c.perm == 3) {
            for (uint64_t r = 0; r < rows; r++) {
                for (int k = 0; k < deg; k++) idx[r * deg + k] = (uint32_t)(sm(s) & (total - 1));
                std::sort(idx.begin() + r * deg, idx.begin() + (r + 1) * deg);
            }
        } else if (c.perm == 4) {
            for (uint64_t r = 0; r < rows; r++) {
                const uint64_t base = (uint64_t)((double)r / (double)rows * (double)(total - c.region));
                for (int k = 0; k < deg; k++) idx[r * deg + k] = (uint32_t)(base + (sm(s) & (c.region - 1)));
            }
        } else {
            for (uint64_t i = 0; i < rows * deg; i++) idx[i] = (uint32_t)(sm(s) & (c.region - 1));
        }
        CK(hipMemcpy(d_idx, idx.data(), rows * deg * 4, hipMemcpyHostToDevice));
        for (int variant = (argc > 2 ? atoi(argv[2]) : 0); variant < 8; variant++) {
            const int unroll = variant;
            float best = 1e9f;
            for (int it = 0; it < 3; it++) {
                CK(hipEventRecord(a));
                dim3 g(2048), b2(256);
                if (variant == 0) hipLaunchKernelGGL((gather_kernel<2, 0>), g, b2, 0, 0, d_idx, regs, out, rows, deg);
                else if (variant == 1) hipLaunchKernelGGL((gather_kernel<4, 0>), g, b2, 0, 0, d_idx, regs, out, rows, deg);
                else if (variant == 2) hipLaunchKernelGGL((gather_kernel<8, 0>), g, b2, 0, 0, d_idx, regs, out, rows, deg);
                else if (variant == 3) hipLaunchKernelGGL((gather_kernel<4, 1>), g, b2, 0, 0, d_idx, regs, out, rows, deg);
                else if (variant == 4) hipLaunchKernelGGL((gather_kernel<4, 2>), g, b2, 0, 0, d_idx, regs, out, rows, deg);
                else if (variant == 5) hipLaunchKernelGGL((gather_kernel<4, 0>), dim3(4096), b2, 0, 0, d_idx, regs, out, rows, deg);
                else if (variant == 6) hipLaunchKernelGGL((gather_kernel<4, 5>), g, b2, 0, 0, d_idx, regs, out, rows, deg);
                else hipLaunchKernelGGL((gather_kernel<4, 6>), g, b2, 0, 0, d_idx, regs, out, rows, deg);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms;
                CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            double gathered = (double)rows * deg * 64, index = (double)rows * deg * 4, wr = (double)rows * 64;
            static const char *vn[] = {"u2", "u4", "u8", "u4-nt", "u4-pair128", "u4-grid4096", "u4-half32", "u4-quadpair128"};
            printf("%-16s %-12s (%d): %8.3f ms  gathered %7.1f GB/s  (+idx+out %7.1f GB/s)  %.2f Ggather/s\n", c.name, vn[variant], unroll,
                   best, gathered / best / 1e6, (gathered + index + wr) / best / 1e6, rows * deg / best / 1e6);
        }
    }
    if (!only[0] || !strcmp(only, "stream")) {
        float best = 1e9f;
        for (int it = 0; it < 3; it++) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, 0, regs, regs + total * 2, total * 2);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("stream copy 4 GiB -> 4 GiB: %.3f ms  %.1f GB/s (read+write)\n", best, 2.0 * total * 32 / best / 1e6);
    }
    return 0;
}
