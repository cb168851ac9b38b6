// TEST INFRASTRUCTURE (tests/test_pool.py): a host-only stand-in for the few HIP runtime calls stract_amd/csrc/hb_pool.h makes, so
// that the caching allocator's bookkeeping (best fit, split, coalescing, trim, the limit, the out-of-memory retry) runs under
// g++ on a machine without a GPU.  "Device memory" is malloc'ed; the fake device has a capacity and counts the runtime calls.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <map>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };

namespace fake_hip {
struct Device {
    size_t capacity = (size_t)1 << 30;
    size_t in_use = 0;
    long mallocs = 0, frees = 0, syncs = 0;
    std::map<void *, size_t> live;
    int current = 0;               // hipSetDevice / hipGetDevice (the fake devices share one capacity)
    std::map<int, long> syncs_on;  // hipDeviceSynchronize calls per device that was current
    std::map<void *, int> dev_of;  // the device that was current when the block was allocated
};
inline Device &dev()
{
    static Device d;
    return d;
}
} // namespace fake_hip

inline hipError_t hipMalloc(void **p, size_t n)
{
    fake_hip::Device &d = fake_hip::dev();
    if (d.in_use + n > d.capacity) {
        *p = nullptr;
        return hipErrorOutOfMemory;
    }
    // (the bytes are never touched by the pool: a small real allocation keeps distinct, ordered, fake addresses cheap)
    static char *next = (char *)((size_t)1 << 40);
    *p = next;
    next += (n + 4095) / 4096 * 4096 + 4096;
    d.live[*p] = n;
    d.dev_of[*p] = d.current;
    d.in_use += n;
    d.mallocs++;
    return hipSuccess;
}
inline hipError_t hipFree(void *p)
{
    fake_hip::Device &d = fake_hip::dev();
    auto it = d.live.find(p);
    if (it == d.live.end()) return hipErrorInvalidValue;
    d.in_use -= it->second;
    d.live.erase(it);
    d.frees++;
    return hipSuccess;
}
inline hipError_t hipGetDevice(int *dv)
{
    *dv = fake_hip::dev().current;
    return hipSuccess;
}
inline hipError_t hipSetDevice(int dv)
{
    fake_hip::dev().current = dv;
    return hipSuccess;
}
inline hipError_t hipDeviceSynchronize()
{
    fake_hip::dev().syncs++;
    fake_hip::dev().syncs_on[fake_hip::dev().current]++;
    return hipSuccess;
}
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b)
{
    *total_b = fake_hip::dev().capacity;
    *free_b = fake_hip::dev().capacity - fake_hip::dev().in_use;
    return hipSuccess;
}
