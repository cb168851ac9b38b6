// hb_guard_alloc.h - debug allocators (build with -DHB_GUARD_ALLOC=<mode>; never in the shipped library:
// `make guard` / `make redzone` / `make poison` build stract_amd/lib/libhyperball_<mode>.so, HB_LIB_PATH selects it).
//
//   mode 1 "guard pages"  every device allocation gets its own virtual-memory mapping that ENDS where the buffer ends
//                         (rounded up to HB_GUARD_ALIGN bytes, default 16), followed by a reserved but UNMAPPED range: a
//                         kernel that reads or writes past the end of a buffer raises a GPU memory access fault at once
//                         instead of depending on what the allocator happened to place there.
//                         With HB_GUARD_COPIES (default 1 in this mode) hipMemcpy* / hipMemset* are replaced too: the
//                         HIP runtime's own copy / fill paths are not trusted with pointers INSIDE a hipMemMap'ed range
//                         (round 3 saw wrong ingest results under this allocator; tools/guard_selftest.hip is the
//                         repro that tells runtime behaviour from library defects).  Device-to-device copies and fills
//                         become kernels; host<->device copies go through a plain hipMalloc bounce buffer + a kernel.
//   mode 2 "poison"       ordinary hipMalloc; fresh memory is filled with 0xA5, freed memory with 0x5A.
//   mode 4 "plain"       the ordinary allocator; the build only gains HB_GUARD_TRACE (and -DHB_DEBUG_BOUNDS: `make bounds`)
//   mode 3 "red zones"    ordinary hipMalloc, over-allocated by HB_GUARD_REDZONE bytes (default 4096) on either side,
//                         the zones filled with 0xC3; hbguard::check_all() (called by the library at the end of every
//                         C-ABI entry point and at every hipFree) verifies all zones with ONE kernel and aborts with the
//                         offending allocation.  Fast enough for the whole GPU test-suite and BASELINE-size runs; sees
//                         stray WRITES next to a buffer (either side), not reads.
//
// Include FIRST in a translation unit (before rocPRIM, so that its internal hipMemsetAsync / hipMemcpyAsync calls are
// replaced as well); it includes the HIP runtime header itself and then defines the macros.
#pragma once
#ifdef HB_GUARD_ALLOC
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#ifndef HB_GUARD_ALIGN
#define HB_GUARD_ALIGN 16
#endif
#ifndef HB_GUARD_REDZONE
#define HB_GUARD_REDZONE 4096
#endif
#ifndef HB_GUARD_COPIES
#define HB_GUARD_COPIES (HB_GUARD_ALLOC == 1)
#endif

namespace hbguard {
struct Rec {
    void *va;
    size_t reserved, mapped, bytes;
    hipMemGenericAllocationHandle_t h;
};
inline std::mutex &mu()
{
    static std::mutex m;
    return m;
}
inline std::map<void *, Rec> &recs()
{
    static std::map<void *, Rec> r;
    return r;
}
// is p inside one of this allocator's buffers?
inline bool owns(const void *p)
{
    std::lock_guard<std::mutex> g(mu());
    auto &r = recs();
    auto it = r.upper_bound((void *)p);
    if (it == r.begin()) return false;
    --it;
    return (const char *)p < (const char *)it->first + it->second.bytes + HB_GUARD_ALIGN;
}

// ---- kernels used by the replacement copies / fills and by the red-zone check --------------------------------------
static __global__ __launch_bounds__(256) void copy_bytes_kernel(uint8_t *dst, const uint8_t *src, size_t n)
{
    // 16-byte body when both pointers allow it, bytes otherwise (debug path: clarity over speed)
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
        const size_t nv = n / 16;
        for (size_t i = tid; i < nv; i += nth) ((uint4 *)dst)[i] = ((const uint4 *)src)[i];
        for (size_t i = nv * 16 + tid; i < n; i += nth) dst[i] = src[i];
    } else {
        for (size_t i = tid; i < n; i += nth) dst[i] = src[i];
    }
}
static __global__ __launch_bounds__(256) void fill_bytes_kernel(uint8_t *dst, uint8_t v, size_t n)
{
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    const size_t head = std::min<size_t>(n, (16 - ((uintptr_t)dst & 15)) & 15);
    for (size_t i = tid; i < head; i += nth) dst[i] = v;
    const uint32_t w = 0x01010101u * v;
    const size_t nv = (n - head) / 16;
    uint4 *body = (uint4 *)(dst + head);
    for (size_t i = tid; i < nv; i += nth) body[i] = make_uint4(w, w, w, w);
    for (size_t i = head + nv * 16 + tid; i < n; i += nth) dst[i] = v;
}
inline unsigned grid_for_bytes(size_t n) { return (unsigned)std::min<size_t>(std::max<size_t>((n / 16 + 255) / 256, 1), 16384); }

inline hipError_t copy_kernel(void *dst, const void *src, size_t n, hipStream_t s)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(copy_bytes_kernel, dim3(grid_for_bytes(n)), dim3(256), 0, s, (uint8_t *)dst, (const uint8_t *)src, n);
    return hipGetLastError();
}
inline hipError_t fill_kernel(void *dst, int v, size_t n, hipStream_t s)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(fill_bytes_kernel, dim3(grid_for_bytes(n)), dim3(256), 0, s, (uint8_t *)dst, (uint8_t)v, n);
    return hipGetLastError();
}

#if HB_GUARD_ALLOC == 2
// ---- poison mode ----------------------------------------------------------------------------------------------------
inline std::map<void *, size_t> &sizes()
{
    static std::map<void *, size_t> r;
    return r;
}
inline hipError_t gmalloc(void **out, size_t bytes)
{
    void *p = nullptr;
    hipError_t e = (hipMalloc)(&p, bytes ? bytes : 1);
    if (e != hipSuccess) return e;
    (void)(hipMemset)(p, 0xA5, bytes);
    (void)hipDeviceSynchronize();
    {
        std::lock_guard<std::mutex> g(mu());
        sizes()[p] = bytes;
    }
    *out = p;
    return hipSuccess;
}
inline hipError_t gfree(void *p)
{
    if (!p) return hipSuccess;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> g(mu());
        auto it = sizes().find(p);
        if (it != sizes().end()) {
            bytes = it->second;
            sizes().erase(it);
        }
    }
    (void)hipDeviceSynchronize();
    if (bytes) (void)(hipMemset)(p, 0x5A, bytes);
    (void)hipDeviceSynchronize();
    return (hipFree)(p);
}
inline void check_all(const char *) {}

#elif HB_GUARD_ALLOC == 3
// ---- red-zone mode --------------------------------------------------------------------------------------------------
struct Zone {
    const uint8_t *base; // start of the front zone
    size_t bytes;        // user bytes between the zones
};
static __global__ __launch_bounds__(256) void check_zones_kernel(const Zone *zones, unsigned count, unsigned long long *bad)
{
    // block b checks allocation b: both zones, 16 bytes per thread step
    const Zone z = zones[blockIdx.x];
    const uint4 *front = (const uint4 *)z.base;
    const size_t back_off = (HB_GUARD_REDZONE + z.bytes + 15) & ~(size_t)15; // the back zone starts 16-byte aligned
    const uint4 *back = (const uint4 *)(z.base + back_off);
    const size_t back_words = (HB_GUARD_REDZONE - 16) / 16; // conservatively inside the allocation
    for (size_t i = threadIdx.x; i < HB_GUARD_REDZONE / 16 + back_words; i += 256) {
        const bool is_back = i >= HB_GUARD_REDZONE / 16;
        const uint4 v = is_back ? back[i - HB_GUARD_REDZONE / 16] : front[i];
        if (v.x != 0xC3C3C3C3u || v.y != 0xC3C3C3C3u || v.z != 0xC3C3C3C3u || v.w != 0xC3C3C3C3u) {
            // record (allocation index, side, first damaged 16-byte word)
            atomicMin(&bad[0], ((unsigned long long)blockIdx.x << 32) | ((unsigned long long)is_back << 31) |
                                   (unsigned long long)(is_back ? i - HB_GUARD_REDZONE / 16 : i));
        }
    }
    (void)count;
}
inline hipError_t gmalloc(void **out, size_t bytes)
{
    const size_t need = ((bytes ? bytes : 1) + 15) / 16 * 16 + 2 * (size_t)HB_GUARD_REDZONE;
    void *raw = nullptr;
    hipError_t e = (hipMalloc)(&raw, need);
    if (e != hipSuccess) return e;
    (void)(hipMemset)(raw, 0xC3, HB_GUARD_REDZONE);
    const size_t back_off = (HB_GUARD_REDZONE + (bytes ? bytes : 1) + 15) & ~(size_t)15;
    (void)(hipMemset)((char *)raw + back_off, 0xC3, need - back_off);
    // the few padding bytes between the end of the user range and the 16-byte aligned back zone carry the pattern too,
    // but are not checked (the kernel reads 16-byte words)
    void *p = (char *)raw + HB_GUARD_REDZONE;
    {
        std::lock_guard<std::mutex> g(mu());
        recs()[p] = Rec{raw, need, need, bytes ? bytes : 1, {}};
    }
    *out = p;
    return hipSuccess;
}
// verifies every live allocation's zones; prints and aborts on damage
inline void check_all(const char *where)
{
    std::vector<Zone> zones;
    std::vector<void *> owners;
    {
        std::lock_guard<std::mutex> g(mu());
        for (auto &kv : recs()) {
            zones.push_back(Zone{(const uint8_t *)kv.second.va, kv.second.bytes});
            owners.push_back(kv.first);
        }
    }
    if (zones.empty()) return;
    static Zone *d_zones = nullptr;
    static unsigned long long *d_bad = nullptr;
    static size_t cap = 0;
    if (zones.size() > cap) {
        if (d_zones) (void)(hipFree)(d_zones);
        cap = zones.size() * 2 + 64;
        if ((hipMalloc)((void **)&d_zones, cap * sizeof(Zone)) != hipSuccess) return;
    }
    if (!d_bad && (hipMalloc)((void **)&d_bad, 8) != hipSuccess) return;
    (void)hipDeviceSynchronize();
    (void)(hipMemcpy)(d_zones, zones.data(), zones.size() * sizeof(Zone), hipMemcpyHostToDevice);
    (void)(hipMemset)(d_bad, 0xFF, 8);
    hipLaunchKernelGGL(check_zones_kernel, dim3((unsigned)zones.size()), dim3(256), 0, nullptr, (const Zone *)d_zones, (unsigned)zones.size(), d_bad);
    unsigned long long bad = ~0ull;
    (void)(hipMemcpy)(&bad, d_bad, 8, hipMemcpyDeviceToHost);
    if (bad != ~0ull) {
        const unsigned idx = (unsigned)(bad >> 32);
        const bool back = (bad >> 31) & 1;
        std::fprintf(stderr, "[hbguard] RED ZONE DAMAGED at %s: allocation %p (%zu bytes), %s zone, 16-byte word %llu\n", where,
                     idx < owners.size() ? owners[idx] : nullptr, idx < zones.size() ? zones[idx].bytes : 0, back ? "back" : "front",
                     bad & 0x7FFFFFFFull);
        std::fflush(stderr);
        std::abort();
    }
}
inline hipError_t gfree(void *p)
{
    if (!p) return hipSuccess;
    Rec r{};
    {
        std::lock_guard<std::mutex> g(mu());
        auto it = recs().find(p);
        if (it == recs().end()) return (hipFree)(p); // not ours (parenthesised: the macro below does not apply)
        r = it->second;
    }
    check_all("hipFree"); // while the record is still listed
    {
        std::lock_guard<std::mutex> g(mu());
        recs().erase(p);
    }
    return (hipFree)(r.va);
}

#elif HB_GUARD_ALLOC == 4
// ---- mode 4: the ordinary allocator (this build only adds HB_GUARD_TRACE and whatever -D the Makefile target gives it) ----
inline hipError_t gmalloc(void **out, size_t bytes) { return (hipMalloc)(out, bytes); }
inline hipError_t gfree(void *p) { return (hipFree)(p); }
inline void check_all(const char *) {}

#else
// ---- guard-page mode ------------------------------------------------------------------------------------------------
inline hipError_t gmalloc(void **out, size_t bytes)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess || !gran) return e != hipSuccess ? e : hipErrorUnknown;
    // Bisection aid: allocations are numbered in call order; HB_GUARD_STRICT_FROM / HB_GUARD_STRICT_TO (default: all) select
    // the half-open range that gets the strict treatment (guard right behind the 16-byte aligned end, pattern fill); the
    // others are "loose" like hipMalloc memory: end rounded up to 256 bytes, zero filled.  HB_GUARD_TRACE lists them.
    static std::atomic<long> seq_counter{0};
    static const long strict_from = std::getenv("HB_GUARD_STRICT_FROM") ? std::atol(std::getenv("HB_GUARD_STRICT_FROM")) : 0;
    static const long strict_to = std::getenv("HB_GUARD_STRICT_TO") ? std::atol(std::getenv("HB_GUARD_STRICT_TO")) : (1L << 62);
    const long seq = seq_counter.fetch_add(1);
    const bool strict = seq >= strict_from && seq < strict_to;
    if (std::getenv("HB_GUARD_TRACE")) std::fprintf(stderr, "[hbguard] alloc #%ld: %zu bytes (%s)\n", seq, bytes, strict ? "strict" : "loose");
    const size_t align = strict ? (size_t)HB_GUARD_ALIGN : (size_t)256;
    size_t need = (bytes + align - 1) / align * align;
    if (!need) need = align;
    const size_t mapped = (need + gran - 1) / gran * gran, reserved = mapped + gran;
    void *va = nullptr;
    if ((e = hipMemAddressReserve(&va, reserved, gran, nullptr, 0)) != hipSuccess) return hipErrorOutOfMemory;
    hipMemGenericAllocationHandle_t h;
    if ((e = hipMemCreate(&h, mapped, &prop, 0)) != hipSuccess) {
        (void)hipMemAddressFree(va, reserved);
        return hipErrorOutOfMemory;
    }
    hipMemAccessDesc ad{};
    ad.location = prop.location;
    ad.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemMap(va, mapped, 0, h, 0)) != hipSuccess || (e = hipMemSetAccess(va, mapped, &ad, 1)) != hipSuccess) {
        (void)hipMemRelease(h);
        (void)hipMemAddressFree(va, reserved);
        return e;
    }
    void *p = (char *)va + (mapped - need);
    {
        std::lock_guard<std::mutex> g(mu());
        recs()[p] = Rec{va, reserved, mapped, bytes, h};
    }
    {
        // fresh memory carries a pattern (default 0xA5; HB_GUARD_FILL=<hex byte>, e.g. 00): a result that depends on what a
        // new allocation happens to hold becomes reproducible instead of allocator-dependent
        static const int fill = [] {
            const char *e = std::getenv("HB_GUARD_FILL");
            return e ? (int)std::strtol(e, nullptr, 16) & 0xFF : 0xA5;
        }();
        (void)fill_kernel(va, strict ? fill : 0, mapped, nullptr);
        (void)hipDeviceSynchronize();
    }
    *out = p;
    return hipSuccess;
}
inline hipError_t gfree(void *p)
{
    if (!p) return hipSuccess;
    Rec r{};
    {
        std::lock_guard<std::mutex> g(mu());
        auto it = recs().find(p);
        if (it == recs().end()) return (hipFree)(p); // not ours (parenthesised: the macro below does not apply)
        r = it->second;
        recs().erase(it);
    }
    (void)hipDeviceSynchronize(); // hipFree's implicit synchronisation
    (void)hipMemUnmap(r.va, r.mapped);
    (void)hipMemRelease(r.h);
    return hipMemAddressFree(r.va, r.reserved);
}
inline void check_all(const char *) {}
#endif

#if HB_GUARD_COPIES
// ---- replacement copies / fills: nothing but kernels ever touches a guarded buffer ---------------------------------
inline void *bounce(size_t *cap_out)
{
    static void *buf = nullptr;
    static const size_t cap = 64u << 20;
    if (!buf && (hipMalloc)(&buf, cap) != hipSuccess) buf = nullptr;
    *cap_out = cap;
    return buf;
}
inline hipError_t gmemset_async(void *dst, int v, size_t n, hipStream_t s = nullptr) { return fill_kernel(dst, v, n, s); }
inline hipError_t gmemset(void *dst, int v, size_t n)
{
    hipError_t e = fill_kernel(dst, v, n, nullptr);
    return e != hipSuccess ? e : hipDeviceSynchronize();
}
inline hipError_t gmemcpy_async(void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t s = nullptr)
{
    if (!n) return hipSuccess;
    const bool dst_dev = owns(dst), src_dev = owns(src);
    if (kind == hipMemcpyDeviceToDevice || (dst_dev && src_dev)) return copy_kernel(dst, src, n, s);
    if (!dst_dev && !src_dev) return (hipMemcpyAsync)(dst, src, n, kind, s); // neither side is a guarded buffer
    size_t cap = 0;
    char *b = (char *)bounce(&cap);
    if (!b) return hipErrorOutOfMemory;
    // the bounce buffer is shared: chunks are serialised on the host (debug path)
    for (size_t off = 0; off < n; off += cap) {
        const size_t k = std::min(cap, n - off);
        hipError_t e;
        if (dst_dev) { // host (or plain device memory) -> guarded buffer
            if ((e = (hipMemcpyAsync)(b, (const char *)src + off, k, hipMemcpyDefault, s)) != hipSuccess) return e;
            if ((e = copy_kernel((char *)dst + off, b, k, s)) != hipSuccess) return e;
        } else { // guarded buffer -> host (or plain device memory)
            if ((e = copy_kernel(b, (const char *)src + off, k, s)) != hipSuccess) return e;
            if ((e = (hipMemcpyAsync)((char *)dst + off, b, k, hipMemcpyDefault, s)) != hipSuccess) return e;
        }
        if ((e = hipStreamSynchronize(s)) != hipSuccess) return e; // the next chunk (or call) reuses the bounce buffer
    }
    return hipSuccess;
}
inline hipError_t gmemcpy(void *dst, const void *src, size_t n, hipMemcpyKind kind)
{
    hipError_t e = gmemcpy_async(dst, src, n, kind, nullptr);
    return e != hipSuccess ? e : hipDeviceSynchronize();
}
#endif
} // namespace hbguard

#define hipMalloc(p, n) hbguard::gmalloc((void **)(p), (size_t)(n))
#define hipFree(p) hbguard::gfree((void *)(p))
#if HB_GUARD_COPIES
#define hipMemcpyAsync(...) hbguard::gmemcpy_async(__VA_ARGS__)
#define hipMemcpy(...) hbguard::gmemcpy(__VA_ARGS__)
#define hipMemsetAsync(...) hbguard::gmemset_async(__VA_ARGS__)
#define hipMemset(...) hbguard::gmemset(__VA_ARGS__)
#endif
#define HB_GUARD_CHECK(where) hbguard::check_all(where)
// HB_GUARD_TRACE=1 in the environment: every kernel launch of the library is named on stderr and followed by a stream
// synchronisation, so the last "launch" line without its "done" names the kernel that faulted (rocPRIM's own launches are
// not traced: a fault between two traced launches is theirs, or a copy's)
namespace hbguard {
inline bool trace()
{
    static const bool on = std::getenv("HB_GUARD_TRACE") != nullptr;
    return on;
}
} // namespace hbguard
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                                     \
    do {                                                                                                                      \
        if (hbguard::trace()) {                                                                                               \
            std::fprintf(stderr, "[hbguard] launch %s  (%s:%d)\n", #kernelName, __FILE__, __LINE__);                          \
            std::fflush(stderr);                                                                                              \
        }                                                                                                                     \
        hipLaunchKernelGGLInternal((kernelName), (numBlocks), (numThreads), (memPerBlock), (streamId), __VA_ARGS__);          \
        if (hbguard::trace()) {                                                                                               \
            const hipError_t e_trace_ = hipStreamSynchronize(streamId);                                                       \
            std::fprintf(stderr, "[hbguard]   done %s: %s\n", #kernelName, hipGetErrorString(e_trace_));                      \
            std::fflush(stderr);                                                                                              \
        }                                                                                                                     \
    } while (0)
#else
#define HB_GUARD_CHECK(where) ((void)0)
#endif
