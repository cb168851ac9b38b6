// TEST INFRASTRUCTURE (tests/test_pool.py): exercises stract_amd/csrc/hb_pool.h - the caching allocator under every hipMalloc /
// hipFree of the library - against tests/fake_hip (a 1 GiB fake device).  Prints "ok" or the line of the first failed check.
#include "hb_pool.h"

#include <cstdio>
#include <vector>

#define CHECK(c)                                                      \
    do {                                                              \
        if (!(c)) {                                                   \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                 \
        }                                                             \
    } while (0)

int main()
{
    const size_t MB = (size_t)1 << 20;
    fake_hip::Device &d = fake_hip::dev();
    hb::DevPool &pool = hb::DevPool::get();
    void *a = nullptr, *b = nullptr, *c = nullptr, *e = nullptr;
    // (limit = half of the fake device = 512 MiB)
    CHECK(hipMalloc(&a, 100 * MB) == hipSuccess && a && d.mallocs == 1 && pool.reserved() == 100 * MB);
    CHECK(hipFree(a) == hipSuccess && d.frees == 0 && d.syncs == 1);       // kept, and synchronised like hipFree would
    CHECK(pool.cached_free() == 100 * MB);
    // reuse with a split: 30 of the cached 100, the rest stays free
    CHECK(hipMalloc(&b, 30 * MB) == hipSuccess && b == a && d.mallocs == 1);
    CHECK(pool.cached_free() == 70 * MB);
    CHECK(hipMalloc(&c, 70 * MB) == hipSuccess && c == (char *)a + 30 * MB && d.mallocs == 1); // exact fit of the rest
    CHECK(pool.cached_free() == 0);
    // double free / interior pointer are errors, not corruption
    CHECK(hipFree((char *)b + 256) == hipErrorInvalidValue);
    CHECK(hipFree(b) == hipSuccess && hipFree(b) == hipErrorInvalidValue);
    // neighbours coalesce: b + c free again = one 100 MiB extent that a 100 MiB request fits
    CHECK(hipFree(c) == hipSuccess && pool.cached_free() == 100 * MB);
    CHECK(hipMalloc(&e, 100 * MB) == hipSuccess && e == a && d.mallocs == 1);
    CHECK(hipFree(e) == hipSuccess);
    // a remainder below 1 MiB is not split off
    CHECK(hipMalloc(&b, 100 * MB - 4096) == hipSuccess && b == a && pool.cached_free() == 0);
    CHECK(hipFree(b) == hipSuccess);
    // best fit: with free extents of 100 and 40, a 35 MiB request takes the 40
    void *small = nullptr, *pin = nullptr;
    CHECK(hipMalloc(&pin, 100 * MB) == hipSuccess && pin == a);            // occupies the 100
    CHECK(hipMalloc(&small, 40 * MB) == hipSuccess && d.mallocs == 2);
    CHECK(hipFree(small) == hipSuccess && hipFree(pin) == hipSuccess);
    CHECK(hipMalloc(&b, 35 * MB) == hipSuccess && b == small);
    CHECK(hipFree(b) == hipSuccess);
    CHECK(pool.reserved() == 140 * MB && pool.peak_reserved() == 140 * MB);
    // trim: entirely free base blocks go back to the runtime, blocks with a live extent stay
    CHECK(hipMalloc(&b, 10 * MB) == hipSuccess);                            // lives in one of the two blocks
    HB_POOL_TRIM();
    CHECK(d.frees == 1 && d.live.size() == 1 && pool.reserved() == (b == small ? 40 : 100) * MB);
    CHECK(hipFree(b) == hipSuccess);
    HB_POOL_TRIM();
    CHECK(d.frees == 2 && d.live.empty() && pool.reserved() == 0 && d.in_use == 0);
    HB_POOL_RESET_PEAK();
    CHECK(HB_POOL_PEAK() == 0);
    // the limit (512 MiB): cached-but-unused blocks are given back before the pool grows past it
    std::vector<void *> v(5);
    for (auto &p : v) CHECK(hipMalloc(&p, 100 * MB) == hipSuccess);
    for (auto &p : v) CHECK(hipFree(p) == hipSuccess);
    CHECK(pool.reserved() == 500 * MB && d.in_use == 500 * MB);
    CHECK(hipMalloc(&b, 150 * MB) == hipSuccess);                           // nothing cached fits; 500 + 150 > 512 -> trim first
    CHECK(pool.reserved() == 150 * MB && d.in_use == 150 * MB && HB_POOL_PEAK() == 500 * MB);
    CHECK(hipFree(b) == hipSuccess);
    HB_POOL_TRIM();
    // out of memory in the runtime: the cache is trimmed and the request tried once more; a hopeless request fails cleanly
    for (auto &p : v) CHECK(hipMalloc(&p, 100 * MB) == hipSuccess);
    CHECK(hipFree(v[0]) == hipSuccess && hipFree(v[1]) == hipSuccess && hipFree(v[2]) == hipSuccess); // 300 cached, 200 live
    d.capacity = 600 * MB;                                                   // (as if someone else took the rest of the device)
    CHECK(hipMalloc(&b, 350 * MB) == hipSuccess && pool.reserved() == 550 * MB); // 500 + 350 > 600: only after the trim
    CHECK(hipMalloc(&c, 100 * MB) == hipErrorOutOfMemory && c == nullptr && pool.reserved() == 550 * MB);
    CHECK(hipFree(b) == hipSuccess && hipFree(v[3]) == hipSuccess && hipFree(v[4]) == hipSuccess);
    HB_POOL_TRIM();
    CHECK(pool.reserved() == 0 && d.in_use == 0 && d.live.empty());
    // the same below the limit: the runtime itself says no (memory held by others), the cache gives way, the retry succeeds
    d.capacity = (size_t)1 << 30;
    CHECK(hipMalloc(&v[0], 120 * MB) == hipSuccess && hipMalloc(&v[1], 100 * MB) == hipSuccess && hipFree(v[0]) == hipSuccess);
    d.capacity = 300 * MB; // 120 cached + 100 live; 150 more do not fit until the 120 go back
    const long frees_before = d.frees;
    CHECK(hipMalloc(&b, 150 * MB) == hipSuccess && d.frees == frees_before + 1 && pool.reserved() == 250 * MB && d.in_use == 250 * MB);
    CHECK(hipFree(b) == hipSuccess && hipFree(v[1]) == hipSuccess);
    HB_POOL_TRIM();
    CHECK(pool.reserved() == 0 && d.live.empty());
    d.capacity = (size_t)1 << 30;
    // a pointer the pool never handed out goes to the runtime's hipFree
    void *foreign = nullptr;
    CHECK((hipMalloc)(&foreign, MB) == hipSuccess && hipFree(foreign) == hipSuccess && d.live.empty());
    // zero bytes is a valid request, a null free a no-op
    CHECK(hipMalloc(&b, 0) == hipSuccess && b && hipFree(b) == hipSuccess && hipFree(nullptr) == hipSuccess);
    // [r5, ADVICE r4] several devices in one process: accounting and limit per device; an extent is released under ITS device's
    // synchronisation whichever device is current, and the current device is left as it was
    HB_POOL_TRIM();
    CHECK(pool.reserved() == 0);
    void *on0 = nullptr, *on1 = nullptr, *again1 = nullptr;
    CHECK(hipSetDevice(0) == hipSuccess && hipMalloc(&on0, 200 * MB) == hipSuccess && pool.reserved() == 200 * MB);
    CHECK(hipSetDevice(1) == hipSuccess && pool.reserved() == 0);                        // device 1 holds nothing yet
    CHECK(hipMalloc(&on1, 300 * MB) == hipSuccess && pool.reserved() == 300 * MB && d.dev_of[on1] == 1);
    CHECK(hipSetDevice(0) == hipSuccess && pool.reserved() == 200 * MB);                // 200 + 300 < 512 would have tripped ONE shared limit at the next block
    const long s0 = d.syncs_on[0], s1 = d.syncs_on[1];
    CHECK(hipFree(on1) == hipSuccess);                                                   // freed while device 0 is current ...
    CHECK(d.syncs_on[1] == s1 + 1 && d.syncs_on[0] == s0 && d.current == 0);             // ... synchronised on device 1, device 0 current again
    CHECK(pool.cached_free() == 0);                                                      // (device 0's view: its only block is live)
    CHECK(hipMalloc(&again1, 300 * MB) == hipSuccess && again1 != on1 && d.dev_of[again1] == 0); // device 0 does not get device 1's extent
    CHECK(hipSetDevice(1) == hipSuccess && pool.cached_free() == 300 * MB);
    HB_POOL_TRIM();                                                                      // trims the CURRENT device only
    CHECK(pool.reserved() == 0 && hipSetDevice(0) == hipSuccess && pool.reserved() == 500 * MB);
    CHECK(hipFree(on0) == hipSuccess && hipFree(again1) == hipSuccess);
    HB_POOL_TRIM();
    CHECK(pool.reserved() == 0 && d.live.empty() && d.in_use == 0);
    std::printf("ok\n");
    return 0;
}
