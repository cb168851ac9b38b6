// hb_guard_alloc.h - debug allocator (build with -DHB_GUARD_ALLOC; never in the shipped library).
// Every device allocation gets its own virtual-memory mapping that ENDS where the buffer ends (rounded up to
// HB_GUARD_ALIGN bytes), followed by a reserved but UNMAPPED range: a kernel or copy that reads or writes past the end of a
// buffer raises a GPU memory access fault at once instead of depending on what the allocator happened to place there.
// -DHB_GUARD_ALLOC=2: poison mode instead (see below).
// Include AFTER every other header of a translation unit: it replaces hipMalloc / hipFree by macros.
#pragma once
#ifdef HB_GUARD_ALLOC
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#ifndef HB_GUARD_ALIGN
#define HB_GUARD_ALIGN 256
#endif
namespace hbguard {
struct Rec {
    void *va;
    size_t reserved, mapped;
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
#if HB_GUARD_ALLOC == 2
// poison mode: ordinary hipMalloc, but fresh memory is filled with 0xA5 and freed memory with 0x5A - code that relies on
// zero-initialised allocations or reads a buffer after freeing it sees garbage instead of plausible data
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
    (void)hipMemset(p, 0xA5, bytes);
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
    if (bytes) (void)hipMemset(p, 0x5A, bytes);
    (void)hipDeviceSynchronize();
    return (hipFree)(p);
}
#else
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
    const size_t align = HB_GUARD_ALIGN;
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
        recs()[p] = Rec{va, reserved, mapped, h};
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
#endif
} // namespace hbguard
#define hipMalloc(p, n) hbguard::gmalloc((void **)(p), (size_t)(n))
#define hipFree(p) hbguard::gfree((void *)(p))
#endif
