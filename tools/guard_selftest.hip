// guard_selftest.hip - is the guard-page debug allocator (stract_amd/csrc/hb_guard_alloc.h, mode 1) itself trustworthy?
//
// Round 3 saw a wrong node set from the record ingest under that allocator and nothing wrong under hipMalloc.  This
// program separates "the HIP runtime / rocPRIM misbehave on hipMemMap'ed memory" from "the library has a defect":
//   part A   the runtime's OWN hipMemcpyAsync (H2D, D2H, D2D) and hipMemsetAsync on pointers at a non-zero offset inside
//            a hipMemMap'ed range, checked by kernels / host compares            (raw calls, parenthesised: no macros)
//   part B   the same operations through the allocator's replacement copies (kernels + bounce buffer)
//   part C   rocPRIM radix_sort_keys(128-bit, double buffer) -> unique -> merge -> unique on guarded buffers, i.e. the
//            node-set loop of hb_ingest.hip, compared with the same pipeline on plain hipMalloc memory and with the
//            host (std::sort / std::unique).  Built twice: -DHB_GUARD_COPIES=0 (rocPRIM's internal hipMemsetAsync /
//            hipMemcpyAsync go to the runtime) and =1 (they are replaced by kernels).
// Prints one line per check; exit code = number of failed checks.
//   hipcc --offload-arch=gfx950 -O2 -DHB_GUARD_ALLOC=1 -DHB_GUARD_COPIES=1 -I stract_amd/csrc tools/guard_selftest.hip -o tools/guard_selftest.bin
#include "hb_guard_alloc.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include <rocprim/rocprim.hpp>

using u128 = rocprim::uint128_t;

static int g_fail = 0;
static void report(const char *what, bool ok, const char *extra = "")
{
    std::printf("%-86s %s %s\n", what, ok ? "ok" : "FAILED", extra);
    std::fflush(stdout);
    if (!ok) g_fail++;
}
#define CK(call)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            std::printf("%s: %s\n", #call, hipGetErrorString(e_));                                 \
            g_fail++;                                                                              \
        }                                                                                          \
    } while (0)

__global__ void pattern_kernel(uint8_t *p, size_t n, uint32_t salt)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (uint8_t)((i * 2654435761u + salt) >> 13);
}
__global__ void verify_kernel(const uint8_t *p, size_t n, uint32_t salt, unsigned long long *bad)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        if (p[i] != (uint8_t)((i * 2654435761u + salt) >> 13)) atomicAdd(bad, 1ull);
}
__global__ void verify_const_kernel(const uint8_t *p, size_t n, uint8_t v, unsigned long long *bad)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        if (p[i] != v) atomicAdd(bad, 1ull);
}

static unsigned long long *d_bad = nullptr; // plain hipMalloc memory
static unsigned long long read_bad()
{
    unsigned long long h = 0;
    (void)(hipMemcpy)(&h, d_bad, 8, hipMemcpyDeviceToHost);
    (void)(hipMemset)(d_bad, 0, 8);
    return h;
}
static uint8_t host_pat(size_t i, uint32_t salt) { return (uint8_t)((i * 2654435761u + salt) >> 13); }

// copies / fills on guarded buffers; raw = the runtime's own entry points
static void copies(bool raw, size_t n)
{
    char label[160];
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, n) != hipSuccess || hipMalloc(&b, n) != hipSuccess) { // macros: guarded
        report("guarded allocation", false);
        return;
    }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const unsigned grid = 2048;
    std::vector<uint8_t> h(n), back(n);
    for (size_t i = 0; i < n; i++) h[i] = host_pat(i, 7);
    // H2D
    if (raw) CK((hipMemcpyAsync)(a, h.data(), n, hipMemcpyHostToDevice, s));
    else CK(hipMemcpyAsync(a, h.data(), n, hipMemcpyHostToDevice, s));
    CK(hipStreamSynchronize(s));
    hipLaunchKernelGGL(verify_kernel, dim3(grid), dim3(256), 0, s, (const uint8_t *)a, n, 7u, d_bad);
    CK(hipStreamSynchronize(s));
    unsigned long long bad = read_bad();
    std::snprintf(label, sizeof(label), "%s H2D of %zu bytes into a guarded buffer (offset %zu in its mapping)", raw ? "runtime" : "replaced", n,
                  (size_t)((uintptr_t)a & ((2u << 20) - 1)));
    report(label, bad == 0);
    // D2D
    hipLaunchKernelGGL(pattern_kernel, dim3(grid), dim3(256), 0, s, (uint8_t *)a, n, 11u);
    hipLaunchKernelGGL(pattern_kernel, dim3(grid), dim3(256), 0, s, (uint8_t *)b, n, 99u);
    if (raw) CK((hipMemcpyAsync)(b, a, n, hipMemcpyDeviceToDevice, s));
    else CK(hipMemcpyAsync(b, a, n, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(verify_kernel, dim3(grid), dim3(256), 0, s, (const uint8_t *)b, n, 11u, d_bad);
    CK(hipStreamSynchronize(s));
    bad = read_bad();
    std::snprintf(label, sizeof(label), "%s D2D of %zu bytes between guarded buffers", raw ? "runtime" : "replaced", n);
    report(label, bad == 0);
    // D2D of a sub-range at an odd offset
    if (n > 4096) {
        hipLaunchKernelGGL(pattern_kernel, dim3(grid), dim3(256), 0, s, (uint8_t *)b, n, 99u);
        const size_t off = 1000, len = n - 3000;
        if (raw) CK((hipMemcpyAsync)((char *)b + off, (char *)a + off, len, hipMemcpyDeviceToDevice, s));
        else CK(hipMemcpyAsync((char *)b + off, (char *)a + off, len, hipMemcpyDeviceToDevice, s));
        CK(hipStreamSynchronize(s));
        CK((hipMemcpy)(back.data(), b, 0, hipMemcpyDeviceToHost)); // no-op; keeps the API path warm
        // check on the device: inside = pattern 11, outside = pattern 99
        std::vector<uint8_t> hb(n);
        if (raw) CK((hipMemcpy)(hb.data(), b, n, hipMemcpyDeviceToHost));
        else CK(hipMemcpy(hb.data(), b, n, hipMemcpyDeviceToHost));
        size_t wrong = 0;
        for (size_t i = 0; i < n; i++) wrong += hb[i] != host_pat(i, (i >= off && i < off + len) ? 11 : 99);
        std::snprintf(label, sizeof(label), "%s D2D of a sub-range (+1000, %zu bytes) and D2H of the whole buffer", raw ? "runtime" : "replaced", len);
        report(label, wrong == 0);
    }
    // memset
    hipLaunchKernelGGL(pattern_kernel, dim3(grid), dim3(256), 0, s, (uint8_t *)a, n, 5u);
    if (raw) CK((hipMemsetAsync)(a, 0x3C, n, s));
    else CK(hipMemsetAsync(a, 0x3C, n, s));
    hipLaunchKernelGGL(verify_const_kernel, dim3(grid), dim3(256), 0, s, (const uint8_t *)a, n, (uint8_t)0x3C, d_bad);
    CK(hipStreamSynchronize(s));
    bad = read_bad();
    std::snprintf(label, sizeof(label), "%s memset of %zu bytes of a guarded buffer", raw ? "runtime" : "replaced", n);
    report(label, bad == 0);
    // D2H
    hipLaunchKernelGGL(pattern_kernel, dim3(grid), dim3(256), 0, s, (uint8_t *)a, n, 21u);
    if (raw) CK((hipMemcpyAsync)(back.data(), a, n, hipMemcpyDeviceToHost, s));
    else CK(hipMemcpyAsync(back.data(), a, n, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    size_t wrong = 0;
    for (size_t i = 0; i < n; i++) wrong += back[i] != host_pat(i, 21);
    std::snprintf(label, sizeof(label), "%s D2H of %zu bytes from a guarded buffer", raw ? "runtime" : "replaced", n);
    report(label, wrong == 0);
    CK(hipStreamDestroy(s));
    (void)hipFree(a);
    (void)hipFree(b);
}

template <class Alloc, class Free>
static bool node_set_pipeline(const std::vector<u128> &c1, const std::vector<u128> &c2, std::vector<u128> *out, Alloc alloc, Free free_)
{
    // hb_ingest.hip's node-set loop for two chunks: sort + unique each, merge, unique
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const size_t n1 = c1.size(), n2 = c2.size(), cap = std::max(n1, n2);
    u128 *keys = (u128 *)alloc(cap * 16), *alt = (u128 *)alloc(cap * 16), *set0 = (u128 *)alloc((n1 + n2) * 16), *set1 = (u128 *)alloc((n1 + n2) * 16);
    uint64_t *d_n = (uint64_t *)alloc(8);
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    auto need = [&](size_t b) {
        if (b <= tmp_bytes) return;
        if (tmp) free_(tmp);
        tmp = alloc(b + 1);
        tmp_bytes = b;
    };
    uint64_t n = 0;
    bool ok = true;
    auto sort_unique = [&](const std::vector<u128> &c, uint64_t *count) {
        CK(hipMemcpyAsync(keys, c.data(), c.size() * 16, hipMemcpyHostToDevice, s));
        rocprim::double_buffer<u128> db(keys, alt);
        size_t bytes = 0;
        CK(rocprim::radix_sort_keys(nullptr, bytes, db, c.size(), 0, 128, s));
        need(bytes);
        CK(rocprim::radix_sort_keys(tmp, bytes, db, c.size(), 0, 128, s));
        u128 *sorted = db.current(), *other = db.alternate();
        bytes = 0;
        CK(rocprim::unique(nullptr, bytes, sorted, other, d_n, c.size(), rocprim::equal_to<u128>(), s));
        need(bytes);
        CK(rocprim::unique(tmp, bytes, sorted, other, d_n, c.size(), rocprim::equal_to<u128>(), s));
        CK(hipMemcpyAsync(count, d_n, 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        if (other != keys) CK(hipMemcpyAsync(keys, other, *count * 16, hipMemcpyDeviceToDevice, s));
    };
    uint64_t cu = 0;
    sort_unique(c1, &cu);
    CK(hipMemcpyAsync(set0, keys, cu * 16, hipMemcpyDeviceToDevice, s));
    n = cu;
    sort_unique(c2, &cu);
    size_t bytes = 0;
    CK(rocprim::merge(nullptr, bytes, set0, keys, set1, (size_t)n, (size_t)cu, rocprim::less<u128>(), s));
    need(bytes);
    CK(rocprim::merge(tmp, bytes, set0, keys, set1, (size_t)n, (size_t)cu, rocprim::less<u128>(), s));
    bytes = 0;
    CK(rocprim::unique(nullptr, bytes, set1, set0, d_n, (size_t)(n + cu), rocprim::equal_to<u128>(), s));
    need(bytes);
    CK(rocprim::unique(tmp, bytes, set1, set0, d_n, (size_t)(n + cu), rocprim::equal_to<u128>(), s));
    CK(hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    if (n > n1 + n2) {
        ok = false;
        n = 0;
    }
    out->resize(n);
    CK(hipMemcpyAsync(out->data(), set0, n * 16, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    CK(hipStreamDestroy(s));
    for (void *p : {(void *)keys, (void *)alt, (void *)set0, (void *)set1, (void *)d_n, tmp})
        if (p) free_(p);
    return ok;
}

__global__ void atomics_kernel(unsigned long long *sum, unsigned int *mx, unsigned int *cas, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        atomicAdd(&sum[i & 63], (unsigned long long)i);
        atomicMax(&mx[i & 63], (unsigned int)(i * 2654435761u));
        atomicCAS(&cas[i & 1023], 0u, (unsigned int)i + 1u);
    }
}
struct IsOdd {
    __device__ bool operator()(uint64_t v) const { return v & 1; }
};

// exclusive_scan, inclusive_scan(max), select(flag iterator), reduce, and global atomics on guarded buffers vs the host
static void scans_and_atomics()
{
    const size_t n = (5u << 20) + 7;
    std::vector<uint64_t> h(n), want_ex(n), want_sel;
    std::mt19937_64 rng(99);
    uint64_t acc = 0, total = 0;
    for (size_t i = 0; i < n; i++) {
        h[i] = rng() & 0xFFFF;
        want_ex[i] = acc;
        acc += h[i];
        if (h[i] & 1) want_sel.push_back(h[i]);
        total += h[i];
    }
    uint64_t *d_in = nullptr, *d_out = nullptr, *d_n = nullptr;
    void *tmp = nullptr;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    CK(hipMalloc((void **)&d_in, n * 8));
    CK(hipMalloc((void **)&d_out, n * 8));
    CK(hipMalloc((void **)&d_n, 8));
    CK(hipMemcpyAsync(d_in, h.data(), n * 8, hipMemcpyHostToDevice, s));
    size_t bytes = 0;
    CK(rocprim::exclusive_scan(nullptr, bytes, d_in, d_out, (uint64_t)0, n, rocprim::plus<uint64_t>(), s));
    CK(hipMalloc(&tmp, bytes + 3));
    CK(rocprim::exclusive_scan(tmp, bytes, d_in, d_out, (uint64_t)0, n, rocprim::plus<uint64_t>(), s));
    std::vector<uint64_t> got(n);
    CK(hipMemcpyAsync(got.data(), d_out, n * 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    report("rocPRIM exclusive_scan on guarded buffers == host", got == want_ex);
    (void)hipFree(tmp);
    bytes = 0;
    CK(rocprim::select(nullptr, bytes, d_in, d_out, d_n, n, IsOdd(), s));
    CK(hipMalloc(&tmp, bytes + 5));
    CK(rocprim::select(tmp, bytes, d_in, d_out, d_n, n, IsOdd(), s));
    uint64_t cnt = 0;
    CK(hipMemcpyAsync(&cnt, d_n, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    got.assign(want_sel.size(), 0);
    CK(hipMemcpyAsync(got.data(), d_out, std::min<size_t>(cnt, want_sel.size()) * 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    report("rocPRIM select on guarded buffers == host", cnt == want_sel.size() && got == want_sel);
    (void)hipFree(tmp);
    bytes = 0;
    CK(rocprim::reduce(nullptr, bytes, d_in, d_n, (uint64_t)0, n, rocprim::plus<uint64_t>(), s));
    CK(hipMalloc(&tmp, bytes + 1));
    CK(rocprim::reduce(tmp, bytes, d_in, d_n, (uint64_t)0, n, rocprim::plus<uint64_t>(), s));
    CK(hipMemcpyAsync(&cnt, d_n, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    report("rocPRIM reduce on guarded buffers == host", cnt == total);
    (void)hipFree(tmp);
    // atomics
    unsigned long long *d_sum = nullptr;
    unsigned int *d_mx = nullptr, *d_cas = nullptr;
    CK(hipMalloc((void **)&d_sum, 64 * 8));
    CK(hipMalloc((void **)&d_mx, 64 * 4));
    CK(hipMalloc((void **)&d_cas, 1024 * 4));
    CK(hipMemsetAsync(d_sum, 0, 64 * 8, s));
    CK(hipMemsetAsync(d_mx, 0, 64 * 4, s));
    CK(hipMemsetAsync(d_cas, 0, 1024 * 4, s));
    const size_t na = 1u << 22;
    hipLaunchKernelGGL(atomics_kernel, dim3(1024), dim3(256), 0, s, d_sum, d_mx, d_cas, na);
    unsigned long long hs[64];
    unsigned int hm[64], hc[1024];
    CK(hipMemcpyAsync(hs, d_sum, sizeof(hs), hipMemcpyDeviceToHost, s));
    CK(hipMemcpyAsync(hm, d_mx, sizeof(hm), hipMemcpyDeviceToHost, s));
    CK(hipMemcpyAsync(hc, d_cas, sizeof(hc), hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    bool ok = true;
    for (int k = 0; k < 64; k++) {
        unsigned long long ws = 0;
        unsigned int wm = 0;
        for (size_t i = k; i < na; i += 64) {
            ws += i;
            wm = std::max(wm, (unsigned int)(i * 2654435761u));
        }
        ok = ok && hs[k] == ws && hm[k] == wm;
    }
    for (int k = 0; k < 1024; k++) ok = ok && hc[k] != 0 && ((hc[k] - 1) & 1023) == (unsigned)k;
    report("global atomicAdd(u64) / atomicMax / atomicCAS on guarded buffers == host", ok);
    for (void *p : {(void *)d_in, (void *)d_out, (void *)d_n, (void *)d_sum, (void *)d_mx, (void *)d_cas}) (void)hipFree(p);
    CK(hipStreamDestroy(s));
}

int main()
{
    CK((hipMalloc)((void **)&d_bad, 8));
    CK((hipMemset)(d_bad, 0, 8));
    std::printf("hb_guard_alloc.h self-test: HB_GUARD_ALLOC=%d HB_GUARD_ALIGN=%d HB_GUARD_COPIES=%d\n", HB_GUARD_ALLOC, HB_GUARD_ALIGN, (int)HB_GUARD_COPIES);
    // sizes that are not multiples of the mapping granularity, so that buffers start inside their first page
    for (size_t n : {(size_t)1000, (size_t)(1u << 20) + 48, (size_t)(300u << 20) + 16}) {
        copies(true, n);
#if HB_GUARD_COPIES
        copies(false, n);
#endif
    }
    // part C
    std::mt19937_64 rng(12345);
    const size_t per = 3u << 20;
    std::vector<u128> c1(per), c2(per);
    for (auto *c : {&c1, &c2})
        for (auto &k : *c) { // ~1/3 duplicates inside a chunk, ~1/2 shared between the chunks
            const uint64_t v = rng() % (per * 2 / 3);
            k = ((u128)(v * 0x9E3779B97F4A7C15ull) << 64) | (u128)(v ^ 0xABCDEFull);
        }
    std::vector<u128> expect(c1);
    expect.insert(expect.end(), c2.begin(), c2.end());
    std::sort(expect.begin(), expect.end());
    expect.erase(std::unique(expect.begin(), expect.end()), expect.end());
    std::vector<u128> got_plain, got_guard;
    bool ok = node_set_pipeline(
        c1, c2, &got_plain,
        [](size_t b) {
            void *p = nullptr;
            (void)(hipMalloc)(&p, b);
            return p;
        },
        [](void *p) { (void)(hipFree)(p); });
    report("rocPRIM sort/unique/merge/unique on plain hipMalloc memory == host result", ok && got_plain == expect);
    ok = node_set_pipeline(
        c1, c2, &got_guard,
        [](size_t b) {
            void *p = nullptr;
            (void)hipMalloc(&p, b);
            return p;
        },
        [](void *p) { (void)hipFree(p); });
    char extra[96];
    std::snprintf(extra, sizeof(extra), "(%zu keys, expected %zu)", got_guard.size(), expect.size());
    report(HB_GUARD_COPIES ? "rocPRIM pipeline on GUARDED buffers, runtime copies/fills replaced by kernels == host result"
                           : "rocPRIM pipeline on GUARDED buffers, runtime copies/fills as they are == host result",
           ok && got_guard == expect, extra);
    scans_and_atomics();
    std::printf("%d check(s) failed\n", g_fail);
    return g_fail;
}
