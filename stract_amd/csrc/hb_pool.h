// hb_pool.h - a caching device allocator under every hipMalloc / hipFree of the library's translation units.
//
// Why (profiles/r04h_ingest_C4_trace.txt): on the MI355X boxes hipMalloc + hipFree cost about 60 ms per GB cycled (the driver
// maps and unmaps the pages), and one load of the 2 B-edge graph cycles ~150 GB through them - record chunks, the endpoint
// table, two 20 GB sort buffers, the planner's work memory, then the resident state: 5 of the 7.9 s of hb_finalize were the
// allocator, not kernels.  The peak of what is LIVE at any time is ~60 GB.  So freed extents are kept and handed out again:
// the sort buffers of the ingest become the planner's sort buffers, the record chunks become the Kahan / size / bitmap
// arrays.  (The planner's own slab heap of rounds 2-3 solved the same problem inside the planner only.)
//
//   pool_malloc   best-fit among the free extents of the cached base blocks (split when the rest is worth keeping), else a new
//                 base block from the runtime; on an out-of-memory the cache is trimmed and the request tried once more
//   pool_free     hipDeviceSynchronize() first - hipFree's implicit synchronisation is part of what callers rely on - then the
//                 extent goes back to its base block's free list (neighbours coalesce)
//   pool_trim     base blocks that are entirely free go back to the runtime (end of a load, hb_destroy): what stays
//                 allocated is what the context really holds (hb_stats.device_bytes)
// One pool per process; the accounting (bytes held, high-water mark, hoarding limit) is kept PER DEVICE [r5, ADVICE r4], and an
// extent is released under its own device's synchronisation whichever device is current.  hb_release_cached_memory()
// (include/hyperball.h) is pool_trim() for callers that share the device with other allocators (RCCL, torch, rocPRIM users),
// which cannot reclaim what this cache holds on their own out-of-memory.
// Debug builds (-DHB_GUARD_ALLOC, hb_guard_alloc.h) and memory-checker builds (-DHB_EXACT_ALLOC: tests/simt under
// AddressSanitizer) bypass it: there every buffer must be its own allocation of its exact size.  Include after
// hb_guard_alloc.h, before any other header of the translation unit.
#pragma once
#include <hip/hip_runtime.h>

#if !defined(HB_GUARD_ALLOC) && !defined(HB_EXACT_ALLOC)
#include <cstdint>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace hb {
class DevPool {
public:
    static DevPool &get()
    {
        static DevPool p;
        return p;
    }
    hipError_t alloc(void **out, size_t bytes)
    {
        if (!out) return hipErrorInvalidValue;
        *out = nullptr;
        const size_t need = round_up(bytes ? bytes : 1, kAlign);
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> g(mu_);
        if (void *p = take(dev, need)) {
            *out = p;
            return hipSuccess;
        }
        // No cached extent fits.  While the device has room the cache may keep growing (its free extents are what makes the
        // next loads cheap); once what it holds plus this request passes the limit - half of the device memory, or
        // HB_POOL_LIMIT_BYTES - the blocks nobody uses go back to the runtime first: the footprint then stays near what is live.
        Dev &d = dev_[dev];
        if (d.reserved + need > limit(d)) trim_locked(dev);
        void *base = nullptr;
        hipError_t e = (hipMalloc)(&base, need);
        if (e != hipSuccess) { // make room: give back what nobody uses, try once more
            (void)hipGetLastError();
            trim_locked(dev);
            e = (hipMalloc)(&base, need);
            if (e != hipSuccess) return e;
        }
        Base b;
        b.ptr = (char *)base;
        b.bytes = need;
        b.dev = dev;
        b.ext[0] = Extent{need, false};
        bases_.push_back(std::move(b));
        d.reserved += need;
        if (d.reserved > d.peak) d.peak = d.reserved;
        *out = base;
        return hipSuccess;
    }
    hipError_t free(void *p)
    {
        if (!p) return hipSuccess;
        // what hipFree does implicitly - nothing in flight ON THE EXTENT'S DEVICE may still use it - happens OUTSIDE the lock (ADVICE r5:
        // a free on device 0 that waits for a long kernel must not block every alloc / free / trim of the other devices): look the
        // extent's device up under the lock, synchronise unlocked, then take the lock again to mark and coalesce.  The extent stays
        // live in between (only its owner frees it), so the second lookup finds it.
        int owner_dev = -1;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (const Base &b : bases_)
                if ((char *)p >= b.ptr && (char *)p < b.ptr + b.bytes) {
                    owner_dev = b.dev;
                    break;
                }
        }
        if (owner_dev < 0) { // not ours (allocated before the pool existed, or by someone else)
            (void)hipDeviceSynchronize();
            return (hipFree)(p);
        }
        {
            int cur = owner_dev;
            (void)hipGetDevice(&cur);
            if (cur != owner_dev) (void)hipSetDevice(owner_dev);
            (void)hipDeviceSynchronize();
            if (cur != owner_dev) (void)hipSetDevice(cur);
        }
        std::lock_guard<std::mutex> g(mu_);
        for (Base &b : bases_) {
            if ((char *)p < b.ptr || (char *)p >= b.ptr + b.bytes) continue;
            const size_t off = (size_t)((char *)p - b.ptr);
            auto it = b.ext.find(off);
            if (it == b.ext.end() || it->second.free) return hipErrorInvalidValue; // not the start of a live extent
            it->second.free = true;
            auto nx = std::next(it);
            if (nx != b.ext.end() && nx->second.free) {
                it->second.size += nx->second.size;
                b.ext.erase(nx);
            }
            if (it != b.ext.begin()) {
                auto pv = std::prev(it);
                if (pv->second.free) {
                    pv->second.size += it->second.size;
                    b.ext.erase(it);
                }
            }
            return hipSuccess;
        }
        return hipErrorInvalidValue; // (the block vanished between the two lookups: a live extent keeps its block, so this is a double free)
    }
    void trim()
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceSynchronize();
        std::lock_guard<std::mutex> g(mu_);
        trim_locked(dev);
    }
    // the CURRENT device's: bytes held from the runtime now / at most since the last reset_peak()
    size_t reserved()
    {
        std::lock_guard<std::mutex> g(mu_);
        return dev_[current()].reserved;
    }
    size_t peak_reserved()
    {
        std::lock_guard<std::mutex> g(mu_);
        return dev_[current()].peak;
    }
    // free bytes inside the current device's cached blocks (what a request could get without asking the runtime)
    size_t cached_free()
    {
        std::lock_guard<std::mutex> g(mu_);
        const int dev = current();
        size_t f = 0;
        for (const Base &b : bases_)
            if (b.dev == dev)
                for (const auto &e : b.ext)
                    if (e.second.free) f += e.second.size;
        return f;
    }
    void reset_peak()
    {
        std::lock_guard<std::mutex> g(mu_);
        Dev &d = dev_[current()];
        d.peak = d.reserved;
    }

private:
    static constexpr size_t kAlign = 256;          // hipMalloc's own guarantee
    static constexpr size_t kMinSplit = 1u << 20;  // a remainder smaller than this stays with the extent
    struct Extent {
        size_t size;
        bool free;
    };
    struct Base {
        char *ptr = nullptr;
        size_t bytes = 0;
        int dev = 0;
        std::map<size_t, Extent> ext; // offset -> extent, covering the block
    };
    struct Dev {
        size_t reserved = 0, peak = 0;
        size_t limit = 0; // 0 = not asked yet (the device must be current when it is)
    };
    static size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
    static int current()
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        return dev;
    }
    // hoarding limit of the device that is current (alloc() runs on the device it allocates for): half of ITS memory
    static size_t limit(Dev &d)
    {
        if (!d.limit) {
            if (const char *e = std::getenv("HB_POOL_LIMIT_BYTES")) d.limit = (size_t)std::strtoull(e, nullptr, 10);
            size_t free_b = 0, total_b = 0;
            if (!d.limit && hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) d.limit = total_b / 2;
            if (!d.limit) d.limit = (size_t)64 << 30;
        }
        return d.limit;
    }
    void *take(int dev, size_t need)
    {
        Base *best_b = nullptr;
        std::map<size_t, Extent>::iterator best_it;
        size_t best = ~(size_t)0;
        for (Base &b : bases_) {
            if (b.dev != dev) continue;
            for (auto it = b.ext.begin(); it != b.ext.end(); ++it)
                if (it->second.free && it->second.size >= need && it->second.size < best) {
                    best = it->second.size;
                    best_b = &b;
                    best_it = it;
                }
        }
        if (!best_b) return nullptr;
        // a request far smaller than the extent would pin a large block for a small buffer: only take it if it is a
        // reasonable share, or the extent can be split
        const size_t off = best_it->first;
        if (best - need >= kMinSplit) {
            best_it->second.size = need;
            best_b->ext[off + need] = Extent{best - need, true};
        }
        best_it->second.free = false;
        return best_b->ptr + off;
    }
    void trim_locked(int dev)
    {
        for (size_t i = 0; i < bases_.size();) {
            Base &b = bases_[i];
            if (b.dev == dev && b.ext.size() == 1 && b.ext.begin()->second.free) {
                (void)(hipFree)(b.ptr);
                dev_[dev].reserved -= b.bytes;
                bases_.erase(bases_.begin() + (long)i);
            } else {
                i++;
            }
        }
    }
    std::mutex mu_;
    std::vector<Base> bases_;
    std::map<int, Dev> dev_;
};
inline hipError_t pool_malloc(void **out, size_t bytes) { return DevPool::get().alloc(out, bytes); }
inline hipError_t pool_free(void *p) { return DevPool::get().free(p); }
inline void pool_trim() { DevPool::get().trim(); }
} // namespace hb

#define hipMalloc(p, n) hb::pool_malloc((void **)(p), (size_t)(n))
#define hipFree(p) hb::pool_free((void *)(p))
#define HB_POOL_TRIM() hb::pool_trim()
#define HB_POOL_RESET_PEAK() hb::DevPool::get().reset_peak()
#define HB_POOL_PEAK() hb::DevPool::get().peak_reserved()
#define HB_POOL_RESERVED() hb::DevPool::get().reserved()
#define HB_POOL_CACHED_FREE() hb::DevPool::get().cached_free()
#else
#define HB_POOL_RESERVED() ((size_t)0)
#define HB_POOL_TRIM() ((void)0)
#define HB_POOL_RESET_PEAK() ((void)0)
#define HB_POOL_PEAK() ((size_t)0)
#define HB_POOL_CACHED_FREE() ((size_t)0)
#endif
